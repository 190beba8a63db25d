"""Drop-in stub for the reference module of the same name (Scripts/task1_unet_plus_plus.py:6).  This path is outside the
accelerated hot path (SURVEY.md section 8f "next" rows); the name is exported so the reference's
app.py star-imports (app.py:7-12) succeed unchanged."""
__all__ = ["holdout_runner_unetplusplus_infection_segmentation"]


def holdout_runner_unetplusplus_infection_segmentation(**kw):
    raise NotImplementedError(
        "holdout_runner_unetplusplus_infection_segmentation: not part of the MI355X U-Net hot path yet (SURVEY.md 8f); "
        "use holdout_runner_unet_infection_segmentation / runner_lung_segmentation")


if __name__ == "__main__":
    holdout_runner_unetplusplus_infection_segmentation()
