import torch
torch.manual_seed(0)
EPS = 1e-16

def ref_forward(X, u0, v0, T, G):
    # reference HALS R=1 (matrix_factorization.py:224-227), grad through last G iterations
    u, v = u0, v0
    for it in range(1, T + 1):
        grad = it >= T - G + 1
        with torch.set_grad_enabled(grad):
            a = X @ v; b = v.T @ v
            u = torch.relu((a + EPS) / (b + EPS))
            a2 = X.T @ u; b2 = u.T @ u
            v = torch.relu((a2 + EPS) / (b2 + EPS))
    return u @ v.T

def gram_fwd_bwd(X, v0, gY, T, G, dt):
    """hand-written forward + backward in the Gram form; X >= 0. returns Y, gX"""
    M, N = X.shape
    eps = torch.tensor(EPS, dtype=dt)
    K = X @ X.T
    s = X.sum(1, keepdim=True)           # (M,1)
    t0 = T - G
    def explicit(vcol):
        a = X @ vcol; b = (vcol * vcol).sum()
        return (a + eps) / (b + eps), b
    def gram(u):
        d = (u * u).sum(); rho = 1 / (d + eps)
        p = K @ u
        a = (p + eps * s) * rho
        nb = (u * p).sum() + 2 * eps * (u * s).sum() + N * eps * eps
        b = nb * rho * rho
        q = 1 / (b + eps)
        return (a + eps) * q, (d, rho, p, nb, q)
    hist = []
    # non-graded warm-up
    u = None
    if t0 > 0:
        u, _ = explicit(v0)
        for i in range(2, t0 + 1):
            u, _ = gram(u)
        d = (u * u).sum()
        vstart = (X.T @ u + eps) / (d + eps)
    else:
        vstart = v0
    u, bstart = explicit(vstart)
    us = [u]
    for i in range(t0 + 2, T + 1):
        un, h = gram(u)
        hist.append((u, h))
        u = un
    uT = u
    dT = (uT * uT).sum(); r = 1 / (dT + eps)
    c = X.T @ uT
    vT = (c + eps) * r
    Y = uT @ vT.T
    # ---- backward
    gu = gY @ vT                      # (M,1)
    gv = gY.T @ uT                    # (N,1)
    gc = gv * r
    gdT = -(gv * vT).sum() * r
    gu = gu + X @ gc + 2 * uT * gdT
    gK = torch.zeros(M, M, dtype=dt); gs = torch.zeros(M, 1, dtype=dt)
    unew = uT
    for (uo, (d, rho, p, nb, q)) in reversed(hist):
        ga = gu * q
        gb = -(gu * unew).sum() * q
        gp = ga * rho + gb * rho * rho * uo
        gs = gs + eps * rho * ga + 2 * eps * rho * rho * gb * uo
        grho = (ga * (p + eps * s)).sum() + 2 * rho * gb * nb
        gd = -rho * rho * grho
        gu = K @ gp + gb * rho * rho * (p + 2 * eps * s) + 2 * uo * gd
        gK = gK + gp @ uo.T + uo @ gp.T       # symmetrised: dL/dK applied as S = gK + gK^T
        unew = uo
    ga1 = gu / (bstart + eps)
    gX = uT @ gc.T + gs @ torch.ones(1, N, dtype=dt) + ga1 @ vstart.T + gK @ X
    return Y, gX

for dt in (torch.float64, torch.float32):
  for (T, G) in ((5, 5), (4, 3), (5, 1), (1, 1), (2, 2), (3,1)):
    for case in ("rand", "zero", "sparse"):
        M, N = 8, 512
        X = torch.rand(M, N, dtype=torch.float64)
        if case == "zero": X = torch.zeros(M, N, dtype=torch.float64)
        if case == "sparse": X = X * (torch.rand(M, N) > 0.7); X[3] = 0; X[:, 5:40] = 0
        u0 = torch.rand(M, 1, dtype=torch.float64); v0 = torch.rand(N, 1, dtype=torch.float64)
        gY = torch.rand(M, N, dtype=torch.float64)
        Xr = X.clone().requires_grad_(True)
        Y = ref_forward(Xr, u0, v0, T, G)
        (gX,) = torch.autograd.grad(Y, Xr, gY)
        Y2, gX2 = gram_fwd_bwd(X.to(dt), v0.to(dt), gY.to(dt), T, G, dt)
        ey = (Y2.double() - Y).abs().max() / max(Y.abs().max(), 1e-300)
        eg = (gX2.double() - gX).abs().max() / max(gX.abs().max(), 1e-300)
        # fp32 reference for comparison
        Xf = X.float().clone().requires_grad_(True)
        Yf = ref_forward(Xf, u0.float(), v0.float(), T, G)
        (gXf,) = torch.autograd.grad(Yf, Xf, gY.float())
        egf = (gXf.double() - gX).abs().max() / max(gX.abs().max(), 1e-300)
        print(str(dt)[6:], T, G, case, "Y %.2e gX %.2e (ref fp32 vs f64: %.2e) max|gX| %.3e" % (ey, eg, egf, gX.abs().max()))
