#!/usr/bin/env python3
"""CPU experiment (NumPy): where does the fp32 error of log_prob (data -> base)
come from?  The flow is evaluated with a selectable dtype per stage
  mlp : conditioner MLP (2->16->16->16)
  nrm : softmax / softplus normalisation of the spline parameters (exp, rcp)
  pos : knot cumsum, bin offset x - x0, output y0 + bh * g(z), base term -x^2/2
  ld  : log-det terms
and compared with the all-float64 evaluation.  D = 2, L = 2, H = 16, K = 5.
Usage: python scripts/numerics/exp_logprob_precision.py [scale] [B]"""
import sys
import numpy as np

K, H, P = 5, 16, 16
LO, HI, MINB, MINS = -10.0, 10.0, 1e-4, 1e-4


def knots(th, dn, dp):
  """th [B,16] (dtype of the MLP stage) -> xk, yk [B,K+1] in dp, slopes [B,K+1] in dn"""
  th = th.astype(dn)
  total = dn((HI - LO) - K * MINB)
  out = []
  for part in range(2):
    u = th[:, part * K:(part + 1) * K]
    e = np.exp(u - u.max(1, keepdims=True))
    w = (e / e.sum(1, keepdims=True) * total + dn(MINB)).astype(dp)
    pos = np.concatenate([np.full((len(u), 1), LO, dp), dp(LO) + np.cumsum(w[:, :-1], 1, dtype=dp),
                          np.full((len(u), 1), HI, dp)], 1)
    out.append(pos)
  off = np.log(np.exp(1.0 - MINS) - 1.0)
  v = th[:, 2 * K:] + dn(off)
  sl = np.maximum(v, 0) + np.log1p(np.exp(-np.abs(v))) + dn(MINS)
  return out[0], out[1], sl.astype(dn)


def rqs_fwd(x, xk, yk, dl, dp, dl_t):
  x = x.astype(dp)
  k = np.clip((x[:, None] >= xk[:, 1:-1]).sum(1), 0, K - 1)
  r = np.arange(len(x))
  x0, x1, y0, y1 = xk[r, k], xk[r, k + 1], yk[r, k], yk[r, k + 1]
  d0, d1 = dl[r, k].astype(dp), dl[r, k + 1].astype(dp)
  bw, bh = x1 - x0, y1 - y0
  s = bh / bw
  z = np.clip((x - x0) / bw, 0, 1)
  z1mz = z - z * z
  st = d1 + d0 - 2 * s
  den = s + st * z1mz
  y = y0 + bh * (s * z * z + d0 * z1mz) / den
  t = dl_t
  ld = 2 * np.log(s.astype(t)) + np.log((d1 * z * z + 2 * s * z1mz + d0 * (1 - z) ** 2).astype(t)) - 2 * np.log(den.astype(t))
  return y, ld.astype(np.float64)


def mlp(w, c, u, dm):
  W0, b0, W1, b1, Wo, bo = [a.astype(dm) for a in w]
  inp = np.stack([np.full(len(u), c, dm), u.astype(dm)], 1)
  h = np.maximum(inp @ W0 + b0, 0)
  h = np.maximum(h @ W1 + b1, 0)
  return h @ Wo + bo


def log_prob(params, y, c, dm, dn, dp, dl_t):
  first = params[:P]
  off = P
  layers = []
  for l in range(2):
    W0 = params[off:off + 32].reshape(2, 16); off += 32
    b0 = params[off:off + 16]; off += 16
    W1 = params[off:off + 256].reshape(16, 16); off += 256
    b1 = params[off:off + 16]; off += 16
    Wo = params[off:off + 256].reshape(16, 16); off += 256
    bo = params[off:off + 16]; off += 16
    layers.append((W0, b0, W1, b1, Wo, bo))
  u = [y[:, 0].astype(dp), y[:, 1].astype(dp)]
  ld_tot = np.zeros(len(y))
  fx, fy, fs = knots(np.broadcast_to(first, (1, P)).astype(np.float64), np.float64, np.float64)   # prepared in f64 ...
  fx, fy = np.repeat(fx.astype(dp), len(y), 0), np.repeat(fy.astype(dp), len(y), 0)              # ... rounded once
  fs = np.repeat(fs.astype(np.float32 if dp == np.float32 else np.float64), len(y), 0)
  for l in (1, 0):
    i0, i1 = (1, 0) if l & 1 else (0, 1)
    o0, ld = rqs_fwd(u[i0], fx, fy, fs, dp, dl_t)
    ld_tot += ld
    th = mlp(layers[l], c, o0, dm)
    xk, yk, sl = knots(th, dn, dp)
    o1, ld = rqs_fwd(u[i1], xk, yk, sl, dp, dl_t)
    ld_tot += ld
    u[i0], u[i1] = o0, o1
  base = (-0.5 * u[0].astype(dp) ** 2 - 0.5 * u[1].astype(dp) ** 2).astype(np.float64) - 2 * 0.9189385332046727
  return base + ld_tot, np.stack(u, 1).astype(np.float64)


def main():
  scale = float(sys.argv[1]) if len(sys.argv) > 1 else 0.2
  B = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
  rng = np.random.default_rng(42)
  params = rng.normal(0, scale, 1200).astype(np.float32).astype(np.float64)
  y = (rng.normal(size=(B, 2)) * 2.0).astype(np.float32).astype(np.float64)
  f32, f64 = np.float32, np.float64
  ref, xr = log_prob(params, y, 0.5, f64, f64, f64, f64)
  print(f"scale {scale} B {B}: |x| max {np.abs(xr).max():.2f}")
  for name, cfg in [("all f32", (f32, f32, f32, f32)),
                    ("mlp f64, rest f32", (f64, f32, f32, f32)),
                    ("mlp f32, rest f64", (f32, f64, f64, f64)),
                    ("mlp+nrm f32, pos+ld f64", (f32, f32, f64, f64)),
                    ("mlp+nrm+ld f32, pos f64", (f32, f32, f64, f32)),
                    ("mlp f32+ld f32, nrm+pos f64", (f32, f64, f64, f32))]:
    lp, x = log_prob(params, y, 0.5, *cfg)
    e = np.abs(lp - ref)
    print(f"  {name:30s} lp err max {e.max():.2e} p99.9 {np.quantile(e, .999):.2e} med {np.median(e):.2e}   x err max {np.abs(x - xr).max():.2e}")


if __name__ == "__main__":
  main()


def sample_f64(params, noise, c):
  """base -> data in float64 (spline inverse), for realistic data points"""
  first = params[:P]
  off = P
  layers = []
  for l in range(2):
    W0 = params[off:off + 32].reshape(2, 16); off += 32
    b0 = params[off:off + 16]; off += 16
    W1 = params[off:off + 256].reshape(16, 16); off += 256
    b1 = params[off:off + 16]; off += 16
    Wo = params[off:off + 256].reshape(16, 16); off += 256
    bo = params[off:off + 16]; off += 16
    layers.append((W0, b0, W1, b1, Wo, bo))
  f64 = np.float64
  u = [noise[:, 0].copy(), noise[:, 1].copy()]
  fx, fy, fs = knots(np.broadcast_to(first, (1, P)).astype(f64), f64, f64)
  n = len(noise)
  fx, fy, fs = np.repeat(fx, n, 0), np.repeat(fy, n, 0), np.repeat(fs, n, 0)

  def inv(y, xk, yk, dl):
    k = np.clip((y[:, None] >= yk[:, 1:-1]).sum(1), 0, K - 1)
    r = np.arange(len(y))
    x0, x1, y0, y1 = xk[r, k], xk[r, k + 1], yk[r, k], yk[r, k + 1]
    d0, d1 = dl[r, k], dl[r, k + 1]
    bw, bh = x1 - x0, y1 - y0
    s = bh / bw
    w = np.clip((y - y0) / bh, 0, 1)
    st = d1 + d0 - 2 * s
    cc = -s * w; b = d0 - st * w; a = s - b
    z = np.clip(-2 * cc / (b + np.sqrt(b * b - 4 * a * cc)), 0, 1)
    return bw * z + x0

  for l in (0, 1):
    i0, i1 = (1, 0) if l & 1 else (0, 1)
    o0 = inv(u[i0], fx, fy, fs)
    th = mlp(layers[l], c, u[i0], f64)
    xk, yk, sl = knots(th, f64, f64)
    o1 = inv(u[i1], xk, yk, sl)
    u[i0], u[i1] = o0, o1
  return np.stack(u, 1)


def main2():
  scale = float(sys.argv[1]) if len(sys.argv) > 1 else 0.2
  B = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
  f32, f64 = np.float32, np.float64
  for seed in (42, 1, 2):
    rng = np.random.default_rng(seed)
    params = rng.normal(0, scale, 1200).astype(f32).astype(f64)
    noise = rng.normal(size=(B, 2)).astype(f32).astype(f64)
    y = sample_f64(params, noise, 0.5).astype(f32).astype(f64)
    ref, xr = log_prob(params, y, 0.5, f64, f64, f64, f64)
    print(f"seed {seed} scale {scale} B {B}: data = own samples; |x| max {np.abs(xr).max():.2f}")
    for name, cfg in [("all f32", (f32, f32, f32, f32)),
                      ("mlp f64, rest f32", (f64, f32, f32, f32)),
                      ("mlp f32, rest f64", (f32, f64, f64, f64)),
                      ("mlp+nrm f32, pos+ld f64", (f32, f32, f64, f64)),
                      ("mlp f32+ld f32, nrm+pos f64", (f32, f64, f64, f32)),
                      ("mlp f64, nrm+pos f64, ld f32", (f64, f64, f64, f32))]:
      lp, x = log_prob(params, y, 0.5, *cfg)
      e = np.abs(lp - ref)
      print(f"  {name:30s} lp err max {e.max():.2e} p99.9 {np.quantile(e, .999):.2e} med {np.median(e):.2e}   x err max {np.abs(x - xr).max():.2e}")


if __name__ == "__main__":
  print("---- data = the flow's own samples (the GPU test's distribution) ----")
  main2()


def main3():
  """does rounding the intermediate layer outputs to fp32 matter? (positions in f64 otherwise)"""
  import types
  f32, f64 = np.float32, np.float64
  g = globals()
  orig = g["rqs_fwd"]
  for seed in (0, 1, 2, 3, 4):
    rng = np.random.default_rng(seed)
    params = rng.normal(0, 0.2, 1200).astype(f32).astype(f64)
    noise = rng.normal(size=(65536, 2)).astype(f32).astype(f64)
    y = sample_f64(params, noise, 0.5).astype(f32).astype(f64)
    ref, xr = log_prob(params, y, 0.5, f64, f64, f64, f64)
    res = {}
    for tag, rnd in (("carry f64", False), ("round layer outputs to f32", True)):
      def wrapped(x, xk, yk, dl, dp, dl_t, _o=orig, _r=rnd):
        yy, ld = _o(x, xk, yk, dl, dp, dl_t)
        return (yy.astype(f32).astype(f64) if _r else yy), ld
      g["rqs_fwd"] = wrapped
      lp, x = log_prob(params, y, 0.5, f32, f64, f64, f32)
      e = np.abs(lp - ref)
      res[tag] = (e.max(), np.quantile(e, .999))
    g["rqs_fwd"] = orig
    print(f"seed {seed}: " + "; ".join(f"{k}: max {a:.2e} p99.9 {b:.2e}" for k, (a, b) in res.items()))


if __name__ == "__main__":
  print("---- intermediate rounding ----")
  main3()


def rqs_fwd_mixed(x, xk, yk, dl, dp, dl_t, refine=True):
  """the GPU's precise path: corner and offset in float64, the in-bin rational in fp32"""
  f32, f64 = np.float32, np.float64
  x = x.astype(f64)
  k = np.clip((x[:, None] >= xk[:, 1:-1]).sum(1), 0, K - 1)
  r = np.arange(len(x))
  x0, x1, y0, y1 = xk[r, k], xk[r, k + 1], yk[r, k], yk[r, k + 1]
  d0, d1 = dl[r, k].astype(f32), dl[r, k + 1].astype(f32)
  bw, bh = (x1 - x0).astype(f32), (y1 - y0).astype(f32)
  dxf = (x - x0).astype(f32)
  if refine:
    s = (bh.astype(f64) / bw.astype(f64)).astype(f32)
    z = np.clip((dxf.astype(f64) / bw.astype(f64)).astype(f32), 0, 1)
  else:
    ibw = (f32(1) / bw)
    s = bh * ibw
    z = np.clip(dxf * ibw, f32(0), f32(1))
  z1mz = z - z * z
  st = d1 + d0 - f32(2) * s
  den = s + st * z1mz
  num = bh * (s * z * z + d0 * z1mz)
  inc = (num.astype(f64) / den.astype(f64)).astype(f32) if refine else num * (f32(1) / den)
  y = y0 + inc.astype(f64)
  ld = f32(2) * np.log(s) + np.log(d1 * z * z + f32(2) * s * z1mz + d0 * (f32(1) - z) ** 2) - f32(2) * np.log(den)
  return y, ld.astype(f64)


def main4():
  f32, f64 = np.float32, np.float64
  g = globals()
  orig = g["rqs_fwd"]
  for seed in (0, 1, 2, 3, 4):
    rng = np.random.default_rng(seed)
    params = rng.normal(0, 0.2, 1200).astype(f32).astype(f64)
    noise = rng.normal(size=(65536, 2)).astype(f32).astype(f64)
    y = sample_f64(params, noise, 0.5).astype(f32).astype(f64)
    ref, xr = log_prob(params, y, 0.5, f64, f64, f64, f64)
    out = []
    for tag, fn in (("in-bin f64", orig), ("in-bin fp32, exact quotients", lambda *a: rqs_fwd_mixed(*a, refine=True)),
                    ("in-bin fp32, rcp", lambda *a: rqs_fwd_mixed(*a, refine=False))):
      g["rqs_fwd"] = fn
      lp, x = log_prob(params, y, 0.5, f32, f64, f64, f32)
      e = np.abs(lp - ref)
      out.append(f"{tag}: max {e.max():.2e} p99.9 {np.quantile(e, .999):.2e}")
    g["rqs_fwd"] = orig
    print(f"seed {seed}: " + "; ".join(out))


if __name__ == "__main__":
  print("---- in-bin evaluation precision ----")
  main4()
