"""Does fz_wgrad write outside its workspace / outputs?  Workspace and outputs sit between sentinel guard zones."""
import ctypes
import sys

import torch

sys.path.insert(0, ".")
from factorizer_amd import _native as N  # noqa: E402

dev = "cuda:0"
S = (40, 48, 40)
V = S[0] * S[1] * S[2]
for dt in (torch.bfloat16, torch.float32):
    for (M, K, qact) in ((128, 256, 2), (128, 128, 0), (256, 128, 0)):
        torch.manual_seed(0)
        p = torch.randn(1, M, *S, device=dev).to(dt)
        q = torch.randn(1, K, *S, device=dev).to(dt)
        d = N.WgradDesc()
        d.p, d.M = p.data_ptr(), M
        d.q[0] = q.data_ptr()
        d.nsrc, d.src_mode, d.c0, d.Cin, d.K, d.Vq = 1, 0, 0, K, K, V
        d.D = d.H = d.W = 0
        d.N, d.Ho, d.Wo = V, 0, 0
        d.qact = qact
        d.B, d.loader, d.accumulate = 1, 0, 0
        d.act_dtype = N.act_dtype(p)
        G = 1 << 20  # guard floats
        SENT = 12345.0
        gw_buf = torch.full((G + M * K + G,), SENT, device=dev)
        gb_buf = torch.full((G + M + G,), SENT, device=dev)
        d.gw = gw_buf.data_ptr() + 4 * G
        d.gbias = gb_buf.data_ptr() + 4 * G
        nb = N.lib().fz_wgrad_workspace_bytes(ctypes.byref(d))
        ws_buf = torch.full((G + nb // 4 + G,), SENT, device=dev)
        rc = N.lib().fz_wgrad(ctypes.byref(d), ws_buf.data_ptr() + 4 * G, N.stream_ptr(p))
        torch.cuda.synchronize()
        assert rc == 0, N.lib().fz_last_error_string()
        bad = {}
        for name, buf, n in (("gw", gw_buf, M * K), ("gbias", gb_buf, M), ("workspace", ws_buf, nb // 4)):
            lo = int((buf[:G] != SENT).sum())
            hi = int((buf[G + n:] != SENT).sum())
            if lo or hi:
                idx_hi = (buf[G + n:] != SENT).nonzero()
                bad[name] = (lo, hi, int(idx_hi.min()) if hi else None, int(idx_hi.max()) if hi else None)
        print(f"{dt} M={M} K={K} qact={qact}: workspace {nb} B; guard violations: {bad if bad else 'none'}")
