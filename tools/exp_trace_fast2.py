"""Runs under rocprofv3 on the GPU box (tools/exp_trace_fast2.sh): batches of the benchmark's frames at a low iniThFAST (argv[1]), so that almost every FAST cell has corners at it —
the second k_fast launch then finds an empty retry list, so its duration is the cost of the early-exit workgroups."""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import orbhip
rng = np.random.default_rng(5)
B, H, W = 512, 480, 752
import bench
imgs = bench.make_batch(32, seed0=0)
frames = torch.from_numpy(np.ascontiguousarray(imgs[np.arange(B) % 32])).cuda()
e = orbhip.ORBextractor(1000, 1.2, 8, int(sys.argv[1]) if len(sys.argv) > 1 else 20, 7)
for _ in range(4):
    k, d, c = e.extract_batch(frames, (0, 0))
torch.cuda.synchronize()
print("mean kp", float(c[:, 0].float().mean()))
