"""The coarse filter's error bound (csrc/flat_collect.hip collect_bounds_kernel), restated in numpy and confronted with an
emulation of what the scan kernel computes: operands rounded to bf16 (round to nearest even), products exact, f32
accumulation in MFMA-sized groups starting from C = beta.  The kernels' exactness only needs E to be an UPPER bound of
|s - s_exact|; this checks the formula (not the device) on data with offsets, large norms and tiny values.  CPU only."""
import numpy as np
import pytest

U = 2.0**-24


def bf16(x):
    """round-to-nearest-even to bfloat16, returned as float32"""
    b = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    r = ((b + 0x7FFF + ((b >> 16) & 1)) >> 16) << 16
    return r.astype(np.uint32).view(np.float32)


def mfma_chain(a, b, c0, group=32):
    """sum_k a_k b_k + c0 with exact products and an f32 running sum updated once per `group` terms (one MFMA each)"""
    acc = np.float32(c0)
    for g in range(0, len(a), group):
        p = (a[g : g + group].astype(np.float64) * b[g : g + group].astype(np.float64)).sum()
        acc = np.float32(np.float64(acc) + p)
    return float(acc)


def flat_bound(metric_l2, x, mu, yn_max, ync_max, d, dyc_max=None):
    """2E of collect_bounds_kernel (same terms, same order).  dyc_max = the largest ||y' - bf16(y')||^2 of the store: the bf16
    rounding term from the ACTUAL residual norms (round 4, cl_bound_mode = 1); None: the worst case 2^-8 per element (mode 0)"""
    xn = float((x.astype(np.float64) ** 2).sum())
    xc = (x - mu).astype(np.float32)
    xnc = float((xc.astype(np.float64) ** 2).sum())
    mun = float((mu.astype(np.float64) ** 2).sum())
    infl = 1.0001
    S = np.sqrt(xn * infl) * np.sqrt(yn_max * infl)
    Sc = np.sqrt(xnc * infl) * np.sqrt(ync_max * infl)
    MY = np.sqrt(mun * infl) * np.sqrt(yn_max * infl)
    al = 2.0 if metric_l2 else 1.0
    bmax = ync_max if metric_l2 else MY
    rnd = al * (2.0**-7 + 2.0**-16) * Sc
    if dyc_max is not None:
        a = np.float32(al) * xc
        dq2 = float(((a - bf16(a)).astype(np.float64) ** 2).sum())
        ndq, ndy = np.sqrt(dq2 * infl), np.sqrt(dyc_max * infl)
        rnd = min(rnd, ndq * np.sqrt(ync_max * infl) + (al * np.sqrt(xnc * infl) + ndq) * ndy)
    es = rnd + 1.25 * (d / 16.0) * 8.0 * U * ((1.0 + 0.0079) * al * Sc + bmax)
    if metric_l2:
        E = es + 4 * U * (xnc + ync_max) + d * U * ync_max + 2 * d * U * S + 4 * U * (xn + yn_max) + 2 * (d + 8) * U * (xn + yn_max)
    else:
        E = es + 4 * U * Sc + 2 * U * MY + d * U * MY + d * U * S
    return 2.0 * E * (1.0 + 2.0**-10) + 8.0 * U * (Sc + MY + xnc + ync_max) + 1e-30


@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("metric_l2", [True, False])
@pytest.mark.parametrize("kind", ["uniform", "offset", "scaled", "tiny", "signed"])
def test_flat_bound_covers_the_emulated_coarse_value(metric_l2, kind, mode):
    seed = {"uniform": 1, "offset": 2, "scaled": 3, "tiny": 4, "signed": 5}[kind] * 2 + int(metric_l2)
    rs = np.random.RandomState(seed)
    d, n, nq = 128, 400, 10
    scale, shift = {"uniform": (1.0, 0.0), "offset": (1.0, 3.0), "scaled": (300.0, 0.0), "tiny": (1e-3, 0.0), "signed": (2.0, -1.0)}[kind]
    xb = (rs.rand(n, d) * scale + shift).astype(np.float32)
    xq = (rs.rand(nq, d) * scale + shift).astype(np.float32)
    mu = xb.mean(0).astype(np.float32)
    yc = (xb - mu).astype(np.float32)  # what the store holds before the bf16 rounding
    yn_max = float((xb.astype(np.float64) ** 2).sum(1).max())
    ync = (yc.astype(np.float64) ** 2).sum(1)
    dyc_max = float((((yc - bf16(yc)).astype(np.float64)) ** 2).sum(1).max()) if mode == 1 else None
    worst = 0.0
    ratio = []
    for q in range(nq):
        x = xq[q]
        E = flat_bound(metric_l2, x, mu, yn_max, float(ync.max()), d, dyc_max) / 2.0
        ratio.append(E / (flat_bound(metric_l2, x, mu, yn_max, float(ync.max()), d) / 2.0))
        xc = (x - mu).astype(np.float32)
        bx = bf16(np.float32(2.0 if metric_l2 else 1.0) * xc)  # alpha rides in the query operand (exact scaling)
        for r in range(0, n, 5):
            by = bf16(yc[r])
            if metric_l2:
                s = mfma_chain(bx, by, np.float32(-ync[r]))  # beta = -||y'||^2
                s_exact = float((xc.astype(np.float64) ** 2).sum()) - float(((x.astype(np.float64) - xb[r]) ** 2).sum())
            else:
                s = mfma_chain(bx, by, np.float32((mu.astype(np.float64) * xb[r]).sum()))  # beta = <mu, y>
                s_exact = float((x.astype(np.float64) * xb[r]).sum()) - float((xc.astype(np.float64) * mu).sum())
            worst = max(worst, abs(s - s_exact) / E)
    assert worst < 0.9, worst  # below E with room: the Cauchy-Schwarz bf16 term dominates and is rarely tight
    assert worst > 1e-4  # ... and the rounding is exercised (not a vacuous check)
    if mode == 1:  # the residual norms of random data sit near 0.4 x 2^-8 of the operand norms: E shrinks ~2x or more
        assert max(ratio) <= 1.0 + 1e-12 and np.mean(ratio) < 0.6, ratio


@pytest.mark.parametrize("metric_l2", [True, False])
def test_flat_bound_holds_at_bf16_rounding_ties(metric_l2):
    """every component sits on a bf16 rounding midpoint (relative error 2^-8 per operand, the worst case) and the
    row is collinear with the query (Cauchy-Schwarz tight).  With the unit roundoff taken as 2^-9 this came out at 1.98 E."""
    d = 128
    t = np.float32(1.0 + 2.0**-8)  # midway between the bf16 neighbours 1 and 1 + 2^-7: rounds to 1 (tie to even)
    xb = np.concatenate([np.full((8, d), t), np.full((8, d), -t)]).astype(np.float32)  # mean row = 0
    mu = xb.mean(0).astype(np.float32)
    assert not mu.any()
    yn_max = float((xb.astype(np.float64) ** 2).sum(1).max())
    worst = 0.0
    dyc_max = float((((xb - bf16(xb)).astype(np.float64)) ** 2).sum(1).max())  # every residual at its worst: 2^-8 |y_i|
    for sign in (1.0, -1.0):
        x = np.full(d, sign * t, dtype=np.float32)
        E = flat_bound(metric_l2, x, mu, yn_max, yn_max, d, dyc_max) / 2.0
        assert E <= flat_bound(metric_l2, x, mu, yn_max, yn_max, d) / 2.0 * (1 + 1e-9)
        bx = bf16(np.float32(2.0 if metric_l2 else 1.0) * x)
        for r in (0, 8):
            by = bf16(xb[r])
            if metric_l2:
                s = mfma_chain(bx, by, np.float32(-yn_max))
                s_exact = float((x.astype(np.float64) ** 2).sum()) - float(((x.astype(np.float64) - xb[r]) ** 2).sum())
            else:
                s = mfma_chain(bx, by, np.float32(0.0))
                s_exact = float((x.astype(np.float64) * xb[r]).sum())
            worst = max(worst, abs(s - s_exact) / E)
    assert 0.9 < worst <= 1.0, worst  # tight (the bound is attained up to its small terms) and not exceeded


def ivf_bound(metric_l2, xn, yn, cn, d, dq2=None, dyn=None):
    """E of ivf_collect_pack_kernel (csrc/ivf_collect.hip): xn = ||x - c||^2 (L2) or ||x||^2 (IP), yn = the list's largest
    ||y - c||^2, cn = ||c||^2; dq2 = ||a - bf16(a)||^2 of the query operand a, dyn = the list's largest ||y' - bf16(y')||^2
    (round 4: the rounding term from the actual residual norms; None = the worst case per element)"""
    infl = 1.0001
    nx, ny, nc = np.sqrt(xn * infl), np.sqrt(yn * infl), np.sqrt(cn * infl)
    S = nx * ny
    al = 2.0 if metric_l2 else 1.0
    rnd = al * (2.0**-7 + 2.0**-16) * S
    if dq2 is not None:
        ndq, ndy = np.sqrt(dq2 * infl), np.sqrt(dyn * infl)
        rnd = min(rnd, ndq * ny + (al * nx + ndq) * ndy)
    if metric_l2:
        return (rnd + 1.25 * (d / 16.0) * 8.0 * U * ((1.0 + 0.0079) * 2.0 * S + xn + yn)
                + (d + 1.0) * U * (xn + yn) + (d + 8.0) * U * (nx + ny) ** 2)
    return (rnd + 1.25 * (d / 16.0) * 8.0 * U * ((1.0 + 0.0079) * S + nx * nc)
            + d * U * nx * nc + U * S + (d + 2.0) * U * nx * (nc + ny))


@pytest.mark.parametrize("metric_l2", [True, False])
def test_ivf_bound_from_residual_norms_covers_random_lists(metric_l2):
    """a list of residual rows around c and queries near it: the emulated coarse value stays inside the residual-norm bound,
    which is well below the worst-case one"""
    rs = np.random.RandomState(11 + int(metric_l2))
    d, n = 128, 300
    c = (rs.rand(d) * 4).astype(np.float32)
    rows = (c + rs.randn(n, d) * 0.3).astype(np.float32)
    yres = (rows - c).astype(np.float32)  # what the store rounds to bf16
    yn = float((yres.astype(np.float64) ** 2).sum(1).max())
    dyn = float(((yres - bf16(yres)).astype(np.float64) ** 2).sum(1).max())
    cn = float((c.astype(np.float64) ** 2).sum())
    worst, ratios = 0.0, []
    for q in range(8):
        x = (c + rs.randn(d) * 0.3).astype(np.float32)
        if metric_l2:
            xr = (x - c).astype(np.float32)
            a = np.float32(2.0) * xr
            xn = float((xr.astype(np.float64) ** 2).sum())
        else:
            a = x
            xn = float((x.astype(np.float64) ** 2).sum())
        dq2 = float(((a - bf16(a)).astype(np.float64) ** 2).sum())
        E = ivf_bound(metric_l2, xn, yn, cn, d, dq2, dyn)
        ratios.append(E / ivf_bound(metric_l2, xn, yn, cn, d))
        for r in range(0, n, 7):
            by = bf16(yres[r])
            if metric_l2:
                b0 = np.float32(-float((yres[r].astype(np.float64) ** 2).sum()))
                g0 = np.float32(-xn)
                s = mfma_chain(bf16(a), by, np.float32(b0 + g0))
                s_exact = -float(((x.astype(np.float64) - rows[r]) ** 2).sum())
            else:
                s = mfma_chain(bf16(a), by, np.float32((x.astype(np.float64) * c).sum()))
                s_exact = float((x.astype(np.float64) * rows[r]).sum())
            worst = max(worst, abs(s - s_exact) / E)
    assert 1e-4 < worst < 0.9, worst
    assert max(ratios) <= 1.0 + 1e-12 and np.mean(ratios) < 0.7, ratios


@pytest.mark.parametrize("metric_l2", [True, False])
def test_ivf_bound_holds_at_bf16_rounding_ties(metric_l2):
    """the same worst case on a residual list: c = 0, rows and query +-(1 + 2^-8) ones"""
    d = 128
    t = np.float32(1.0 + 2.0**-8)
    n2 = float(d) * float(t) ** 2
    E = ivf_bound(metric_l2, n2, n2, 0.0, d)
    worst = 0.0
    for sx in (1.0, -1.0):
        x = np.full(d, sx * t, dtype=np.float32)
        for sy in (1.0, -1.0):
            y = np.full(d, sy * t, dtype=np.float32)
            if metric_l2:  # chain from C = beta + gamma = -||y'||^2 - ||x'||^2: s ~ -||x - y||^2
                s = mfma_chain(bf16(np.float32(2.0) * x), bf16(y), np.float32(np.float32(-n2) + np.float32(-n2)))
                s_exact = -float(((x.astype(np.float64) - y) ** 2).sum())
            else:
                s = mfma_chain(bf16(x), bf16(y), np.float32(0.0))
                s_exact = float((x.astype(np.float64) * y).sum())
            worst = max(worst, abs(s - s_exact) / E)
    assert 0.9 < worst <= 1.0, worst


def test_bf16_emulation_is_round_to_nearest_even():
    x = np.array([1.0, 1.00390625, 1.01171875, -3.0000001, 0.0], dtype=np.float32)
    y = bf16(x)
    assert y[0] == 1.0 and y[1] == 1.0 and y[2] == np.float32(1.015625)  # tie to even, tie to even (up)
    assert np.all(np.abs(y - x) <= np.abs(x) * 2.0**-8)
