"""CPU oracle package -- test infrastructure only (see unet_oracle.py header)."""
