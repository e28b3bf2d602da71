import numpy as np
np.random.seed(0)
# F(6,3) matrices (Lavin): points 0, ±1, ±2, ±1/2, inf
BT = np.array([
 [1, 0, -21/4, 0, 21/4, 0, -1, 0],
 [0, 1, 1, -17/4, -17/4, 1, 1, 0],
 [0, -1, 1, 17/4, -17/4, -1, 1, 0],
 [0, 1/2, 1/4, -5/2, -5/4, 2, 1, 0],
 [0, -1/2, 1/4, 5/2, -5/4, -2, 1, 0],
 [0, 2, 4, -5/2, -5, 1/2, 1, 0],
 [0, -2, 4, 5/2, -5, -1/2, 1, 0],
 [0, -1, 0, 21/4, 0, -21/4, 0, 1]], dtype=np.float64)
G = np.array([
 [1, 0, 0],
 [-2/9, -2/9, -2/9],
 [-2/9, 2/9, -2/9],
 [1/90, 1/45, 2/45],
 [1/90, -1/45, 2/45],
 [32/45, 16/45, 8/45],
 [32/45, -16/45, 8/45],
 [0, 0, 1]], dtype=np.float64)
AT = np.array([
 [1, 1, 1, 1, 1, 1, 1, 0],
 [0, 1, -1, 2, -2, 1/2, -1/2, 0],
 [0, 1, 1, 4, 4, 1/4, 1/4, 0],
 [0, 1, -1, 8, -8, 1/8, -1/8, 0],
 [0, 1, 1, 16, 16, 1/16, 1/16, 0],
 [0, 1, -1, 32, -32, 1/32, -1/32, 1]], dtype=np.float64)
K, N, T = 512, 16, 64
d = np.random.uniform(-1, 1, (T, 8, 8, K))
g = np.random.uniform(-1, 1, (3, 3, K, N)) * 0.05
def direct(d, g):
    y = np.zeros((T, 6, 6, N))
    for a in range(3):
        for b in range(3):
            y += np.einsum('tijk,kn->tijn', d[:, a:a+6, b:b+6, :], g[a, b])
    return y
ref = direct(d, g)
def wino(dt_u, dt_v, dt_m, dt_o, scale=None):
    f = lambda x, t: x.astype(t)
    Gm, Bm, Am = G, BT, AT
    U = np.einsum('ia,abkn,jb->ijkn', f(Gm, dt_u), f(g, dt_u), f(Gm, dt_u)).astype(dt_u)
    # 2-pass with intermediate rounding
    t1 = np.einsum('ia,tabk->tibk', f(Bm, dt_v), f(d, dt_v)).astype(dt_v)
    V = np.einsum('tibk,jb->tijk', t1, f(Bm, dt_v)).astype(dt_v)
    U32, V32 = U.astype(np.float32), V.astype(np.float32)
    if dt_m == np.float32:
        # emulate fp32 accumulation in chunks
        M = np.zeros((T, 8, 8, N), np.float32)
        for k0 in range(0, K, 1):
            M += V32[..., k0:k0+1].astype(np.float32) * U32[None, :, :, k0, :]
    else:
        M = np.einsum('tijk,ijkn->tijn', V32.astype(np.float64), U32.astype(np.float64))
    t2 = np.einsum('ia,tabn->tibn', f(Am, dt_o), f(M, dt_o)).astype(dt_o)
    y = np.einsum('tibn,jb->tijn', t2, f(Am, dt_o)).astype(dt_o)
    return y.astype(np.float64)
def rel(y): return np.abs(y - ref).sum() / np.abs(ref).sum()
f32, f64 = np.float32, np.float64
yd = np.zeros((T,6,6,N), np.float32)
for a in range(3):
    for b in range(3):
        for k in range(K):
            yd += d[:, a:a+6, b:b+6, k:k+1].astype(f32) * g[a, b, k].astype(f32)[None, None, None, :]
print("direct fp32 seq", rel(yd.astype(f64)))
for name, cfg in [("all fp32", (f32,f32,f32,f32)), ("U fp64", (f64,f32,f32,f32)), ("V fp64", (f32,f64,f32,f32)),
                  ("M fp64acc", (f32,f32,f64,f32)), ("O fp64", (f32,f32,f32,f64)), ("U,V fp64", (f64,f64,f32,f32)),
                  ("U,V,O fp64", (f64,f64,f32,f64)), ("all but M fp32->fp64", (f64,f64,f32,f64)), ("only storage fp32", (f64,f64,f64,f64))]:
    print(name, rel(wino(*cfg)))
