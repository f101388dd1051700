"""Stand-alone: the plain grouped weight-gradient launch (no AdamW: its fabric reads are operand fetches only) on the STEP's table of
problems -- or on its long-K / short-K part -- under GSTVD_GROUP_ORDER; for rocprofv3 --pmc FETCH_SIZE runs and event timing.
    python tools/wgrad_table_probe.py [all|long|short|uniform]     (GSTVD_GROUP_ORDER in the environment)
Prints the operand bytes every problem needs once, so that FETCH_SIZE x 2 / that = how many times the launch fetched its operands."""
import sys
import os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gst_visdial_amd import ops          # noqa: E402


def step_table():
    """(M, N, K) = (out features, in features, batch rows) of one 16-row train step's weight gradients, backward order."""
    shapes = [(30528, 768, 400)]
    for _ in range(12):
        shapes += [(768, 3072, 400), (3072, 768, 400), (768, 768, 400), (768, 768, 400), (768, 768, 400), (2304, 768, 400)]
    shapes += [(18432, 768, 4688), (768, 1024, 592), (768, 768, 4096)]
    for _ in range(6):
        shapes += [(768, 3072, 4096), (3072, 768, 4096), (768, 1024, 4096), (3072, 768, 4096)]
        shapes += [(1024, 1024, 592)] * 3 + [(3072, 1024, 592)] + [(1024, 1024, 592)] * 3 + [(3072, 1024, 592)]
        for _ in range(2):
            shapes += [(768, 3072, 4096), (3072, 768, 4096), (768, 768, 4096), (2304, 768, 4096)]
    shapes += [(1024, 2048, 592)]
    return shapes


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    dev = torch.device("cuda:0")
    shapes = step_table()
    if which == "long":
        shapes = [s for s in shapes if s[2] > 2400]
    elif which == "short":
        shapes = [s for s in shapes if s[2] <= 2400]
    elif which == "uniform":
        shapes = [(3072, 768, 4096)] * 56
    grp = ops.GemmGroup(dev)
    keep, once = [], 0
    for i, (M, N, K) in enumerate(shapes):
        A = torch.randn(K, M, device=dev, dtype=torch.bfloat16) * 0.05
        B = torch.randn(K, N, device=dev, dtype=torch.bfloat16)
        Cw = torch.empty(M, N, device=dev, dtype=torch.float32)
        keep.append((A, B, Cw))
        once += 2 * K * (M + N)
    tiles = sum(((M + 255) // 256) * ((N + 255) // 256) for M, N, K in shapes)

    def launch():
        for (A, B, Cw), (M, N, K) in zip(keep, shapes):
            grp.add(A, B, Cw, M, N, K, False)
        grp.flush()
    for _ in range(2):
        launch()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); launch(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    print("table %s, GSTVD_GROUP_ORDER=%s: %d problems, %d tiles, operands once %.3f GB, C written %.3f GB; launch %.1f us"
          % (which, os.environ.get("GSTVD_GROUP_ORDER", "default"), len(shapes), tiles, once / 1e9,
             sum(4 * M * N for M, N, K in shapes) / 1e9, best * 1e3))


if __name__ == "__main__":
    main()
