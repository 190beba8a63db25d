import time
try:
    import amdsmi
    amdsmi.amdsmi_init()
    hs = amdsmi.amdsmi_get_processor_handles()
    print("handles", len(hs))
    h = hs[0]
    t=time.time(); p = amdsmi.amdsmi_get_power_info(h); print("power", p, time.time()-t)
    t=time.time(); c = amdsmi.amdsmi_get_clock_info(h, amdsmi.AmdSmiClkType.GFX); print("clk", c, time.time()-t)
    try:
        t=time.time(); m = amdsmi.amdsmi_get_gpu_metrics_info(h); print({k: m[k] for k in m if 'power' in k or 'gfxclk' in k or 'throttle' in k}, time.time()-t)
    except Exception as e: print("metrics", e)
except Exception as e:
    print("amdsmi failed", repr(e))
