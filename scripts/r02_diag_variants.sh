#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r02; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_timed_configs.py -m gpu -q --timeout 600 -k "kernel_form" 2>&1 | grep -E "FAILED|passed|failed|AssertionError: " | tee $O/pytest_variants_all.log
python - <<'PY' 2>&1 | tee $O/variant8_diag.txt
import numpy as np, torch, sys
sys.path.insert(0, '.')
import dsac_amd
from dsac_amd import synth
from oracle import oracle as orc
H, W = 480, 640; P = H*W
fr = synth.chess_like_frame(H, W, seed=2024); uv = synth.pixel_grid(H, W)
N = 256
poses, sets, ok, _ = orc.sample(N, 9, fr["xyz"], uv, H, W, fr["cam"], thr=10.0, max_tries=1 << 16)
ref = orc.get_diff_maps(poses[:8], fr["xyz"], uv, H, W, fr["cam"])
dev = torch.device("cuda:0")
e = dsac_amd.Engine(0)
e.set_frame(fr["xyz"], None, H, W, fr["cam"])
for v in (4, 8, 16, 24, 27):
    for o in (1, 0):
        e.set_option("k2_variant", v); e.set_option("k2_order", o)
        err = torch.zeros(N, P, dtype=torch.float32, device=dev)
        soft = torch.zeros(N, dtype=torch.float64, device=dev)
        e.reproject(torch.from_numpy(poses).to(dev), N=N, err=err, soft=soft); e.synchronize()
        got = err[:8].cpu().numpy()
        d = np.abs(got - ref); bad = np.argwhere(d > 1e-3)
        print("variant", v, "order", o, "max diff %.3e" % d.max(), "n bad", len(bad), "first bad", bad[:5].tolist(), "bad px mod 256", sorted(set((bad[:, 1] % 256).tolist()))[:20] if len(bad) else "")
PY
