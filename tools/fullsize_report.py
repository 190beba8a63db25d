"""Print every error of the full-size known-answer cases (tests/test_gpu_fullsize.py asserts them) and, with --json FILE, write the per-tensor relative errors
of the parameter gradients -- the measured figures tests/golden/fullsize_measured.json holds:  python tools/fullsize_report.py [--json out.json] [case ...]"""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import test_gpu_fullsize as T
import fullsize_cases as FC

args = sys.argv[1:]
flips = "--flips" in args
if flips:
    args.remove("--flips")
jout = None
if "--json" in args:
    i = args.index("--json"); jout = args[i + 1]; del args[i:i + 2]
res = {}
for name in (args or list(FC.CASES)):
    z, arch, ld, p, g, wa = T._run(name, "fp32")
    ps = p if arch == "classifier" else p[::T.P_STRIDE]
    print(f"== {name}: loss {ld[0]:.8f} vs {float(z['loss']):.8f}, metric {ld[1]:.8f} vs {float(z['metric']):.8f}, max|dp| {np.abs(ps - z['p_sample']).max():.2e}, dmean {abs(p.mean(dtype=np.float64) - float(z['p_mean'])):.2e}")
    res[name] = {}
    for k in [k[6:] for k in z.files if k.startswith("gnorm/")]:
        want = float(z["gnorm/" + k]); got = float(np.linalg.norm(g[k].astype(np.float64)))
        full = f" relerr {T.relerr(g[k], z['grad/' + k]):.2e} cos {T.cosine(g[k], z['grad/' + k]):.6f}" if "grad/" + k in z.files else ""
        res[name][k] = {"norm_rel": abs(got - want) / (want + 1e-30), "fp32ref": float(z["fp32ref_relerr/" + k])}
        if "grad/" + k in z.files:
            res[name][k]["relerr"] = T.relerr(g[k], z["grad/" + k])
        print(f"  {k:20s} norm {got:.6e} want {want:.6e} rel {abs(got - want) / (want + 1e-30):.2e}  E_fp32cpu {float(z['fp32ref_relerr/' + k]):.2e}  dsum {abs(float(g[k].astype(np.float64).sum()) - float(z['gsum/' + k])):.2e}{full}")
    for k in [k[8:] for k in z.files if k.startswith("bn_mean/")]:
        mu = wa[k + "/mean"].astype(np.float64) / 0.01; va = (wa[k + "/var"].astype(np.float64) - 0.99) / 0.01
        print(f"  bn {k:12s} dmean {np.abs(mu - z['bn_mean/' + k]).max():.2e} (max {np.abs(z['bn_mean/' + k]).max():.2e})  dvar {np.abs(va - z['bn_var/' + k]).max():.2e} (max {np.abs(z['bn_var/' + k]).max():.2e})")
    if flips:
        # Where do the gradient errors above come from?  The same step on the STRICT fp32 family (conv_algo = 2: exact fp32 products, another summation order): every
        # ReLU output of both engines compared sign by sign on the device, every parameter gradient engine against engine.  Two fp32 evaluations of one
        # step differ by round-off (~1e-7) except where a pre-activation rounds to the other side of zero: each such flip is a discontinuity of the gradient.
        import torch
        from covidseg_amd.engine import HipUNet
        from covidseg_amd import weights as W
        arch_, size, n, _, _ = FC.CASES[name]
        _, w, x, y = FC.build(name)
        e1 = HipUNet(size, size, 1, arch=arch_, dropout_rate=0.0); e1.set_weights(w); e1.forward_backward(x, y)
        e2 = HipUNet(size, size, 1, arch=arch_, dropout_rate=0.0, conv_algo=2); e2.set_weights(w); e2.forward_backward(x, y)
        g1, g2 = e1.get_grads(), e2.get_grads()
        fl = {}
        for lname, kind, ci, co in W.layer_table(1, arch_, (size, size)):
            if kind != "conv3":
                continue
            try:
                a, b = e1.tap_device(n, lname), e2.tap_device(n, lname)
            except Exception:
                continue                                   # (a tensor one of the programs does not materialise)
            differ = int(((a > 0) != (b > 0)).sum().item()); tot = a.numel()
            fl[lname] = {"sign_flips_h2_vs_strict": differ, "elements": tot, "fraction": differ / tot}
        res[name]["_flips"] = fl
        res[name]["_h2_vs_strict_relerr"] = {k: T.relerr(g1[k], g2[k]) for k in g1}
        print("  ReLU sign disagreements between the h2 and the strict-fp32 engine (same step, same weights):")
        for lname, v in fl.items():
            print(f"    {lname:8s} {v['sign_flips_h2_vs_strict']:8d} of {v['elements']:.3e} ({v['fraction']:.2e})")
        worst = sorted(res[name]["_h2_vs_strict_relerr"].items(), key=lambda kv: -kv[1])[:8]
        print("  gradient distance h2 vs strict fp32 (relative L2), worst tensors:", ", ".join(f"{k} {v:.2e}" for k, v in worst))
        del e1, e2
        torch.cuda.empty_cache()
if jout:
    json.dump(res, open(jout, "w"), indent=1)
