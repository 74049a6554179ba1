import sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
import bench, orbhip
frames = bench.make_batch(512)
d = torch.from_numpy(frames).cuda()
ex = orbhip.ORBextractor(1000, 1.2, 8, 20, 7, device=0, max_batch=512)
ref = None
bad = 0
for it, B in enumerate([512, 64, 512, 128, 64, 96, 512, 65, 512]):
    out = ex.extract_batch(d[:B].contiguous(), (0, 1000))
    torch.cuda.synchronize()
    k, de, c = out[0].cpu().numpy().copy(), out[1].cpu().numpy().copy(), out[2].cpu().numpy().copy()
    if ref is None:
        ref = (k, de, c)
    else:
        for b in range(B):
            n = c[b, 0]
            if n != ref[2][b, 0] or not np.array_equal(de[b, :n], ref[1][b, :n]) or not np.array_equal(k[b, :n].view(np.int32), ref[0][b, :n].view(np.int32)):
                bad += 1
    print(it, B, ex.last_fast_passes(), "mismatching frames so far:", bad)
print("OK" if bad == 0 else "FAIL")
