"""Direct (igemm_conv_kernel) against Winograd (wino_conv_kernel) on the step's 3x3 shapes, kernel time from a rocprofv3
kernel trace.  usage (on the GPU box):
  cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/wino_trace -o t -- \
      python3 $GRAFT_REPO_ROOT/tools/time_wino.py run
  python tools/time_wino.py parse gpurun_out/wino_trace
"""
import csv
import glob
import os
import sys

SHAPES_ALL = [(32, 256, 256, 32, 32), (32, 128, 128, 64, 64), (32, 64, 64, 96, 96), (32, 256, 256, 96, 32), (32, 128, 128, 160, 64),
              (96, 16, 16, 256, 256), (32, 32, 32, 128, 128), (32, 64, 64, 224, 96), (32, 16, 16, 256, 256), (32, 256, 256, 32, 96),
              (96, 64, 64, 64, 64), (96, 32, 32, 128, 128)]
SHAPES = SHAPES_ALL[:3] if os.environ.get("WINO_ONLY") else SHAPES_ALL
REPS = 6


def run():
    import ctypes as C
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch
    from dep_gan_im_amd import _lib
    lib = _lib.load()
    dev = torch.device("cuda:0")
    P = lambda t: C.c_void_p(t.data_ptr())
    for B, H, W, ci, co in SHAPES:
        x = torch.randn(B, H, W, ci, device=dev).relu_()
        w = torch.randn(3, 3, ci, co, device=dev) * 0.05
        b = torch.zeros(co, device=dev)
        out = torch.empty(B, H, W, co, device=dev)
        for path in ((8,) if os.environ.get("WINO_ONLY") else (6, 8)):
            for _ in range(REPS):
                _lib.check(lib.depgan_op_conv2d(P(x), P(w), P(b), P(out), B, H, W, ci, co, 3, 1, path, None))
        torch.cuda.synchronize()


def parse(d):
    f = sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True))[-1]
    rows = [r for r in csv.DictReader(open(f)) if "igemm_conv_kernel" in r["Kernel_Name"] or "wino_" in r["Kernel_Name"]]
    if os.environ.get("WINO_ONLY"):   # ablation runs: the Winograd launches only
        rows = [r for r in rows if "wino_" in r["Kernel_Name"]]
        for i, (B, H, W, ci, co) in enumerate(SHAPES):
            rr = rows[i * REPS:(i + 1) * REPS]
            us = min((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rr[1:])
            print("b%d %dx%d %d->%d: %.1f us" % (B, H, W, ci, co, us))
        return
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    assert len(rows) == len(SHAPES) * 2 * REPS, (len(rows), len(SHAPES) * 2 * REPS)
    tot = [0.0, 0.0]
    for i, (B, H, W, ci, co) in enumerate(SHAPES):
        us = []
        for k in range(2):
            rr = rows[(2 * i + k) * REPS:(2 * i + k + 1) * REPS]
            want = "wino" if k else "igemm_conv_kernel"
            assert all(want in r["Kernel_Name"] for r in rr), [r["Kernel_Name"] for r in rr]
            us.append(min((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rr[1:]))
        fl = 2.0 * B * H * W * ci * co * 9
        print("b%d %dx%d %d->%d: direct %.1f us (%.3f)  winograd %.1f us (%.3f algorithmic, %.3f of issued MFMA)  x%.2f" % (
            B, H, W, ci, co, us[0], fl / us[0] / 1e6 / 157.3, us[1], fl / us[1] / 1e6 / 157.3,
            fl * 4 / 9 / us[1] / 1e6 / 157.3, us[0] / us[1]))
        tot[0] += us[0]
        tot[1] += us[1]
    print("sum: direct %.1f us, winograd %.1f us" % tuple(tot))


if __name__ == "__main__":
    run() if sys.argv[1] == "run" else parse(sys.argv[2])
