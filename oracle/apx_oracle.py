"""apx_oracle.py -- TEST INFRASTRUCTURE: the exact output law of the reference's APX-UGS sampler for k = 3 on small graphs.

Only tests/ may import this (it is the checker of the GPU variant ugs_apx_gpu.hip and of the sequential restatement
ugs_apx.cpp; the product never does).

The reference (src/samplers/apx_ugs_sampler/src/apx_ugs_sampler.cpp) is a rejection sampler: a trial draws a root v with
probability est(v)/Z (:409-419), grows an ordered vertex set S = (v, a, b) (APX-RAND-GROW :243-312), estimates the probability
p_hat of having grown S over both orders of {a, b} (APX-PROB :318-382) and accepts with min(1, (beta/Z) k^-C1 / (est(v) p_hat))
(:440-447); the first accepted trial is the sample.  Hence

    law(S)  proportional to  P_root(v) * P_grow(S | v) * E[accept(S)],

and every factor is a finite expectation over independent binomial counts: EstimateCuts (:175-236) draws h neighbours of u with
replacement and counts those that come after v in the APX-DD order and lie outside the current set -- hits ~ Binomial(h, f) with
f = (such neighbours) / deg(u) -- and reports deg(u) * hits / h if hits >= ell, else 0.  For every parameter set the reference
can reach with k = 3 the formulas of :184-196 give h = 100, ell = 50 (computed below, not assumed).
  * P_grow: step 1 needs the root's estimate to be positive (one binomial) and picks a uniformly among the root's later
    neighbours; step 2 picks the side (v or a) with probability cut/total (expectation over two binomials) and then b uniformly
    among that side's later neighbours outside the set.
  * E[accept]: p_hat = p(v,x,y) + p(v,y,x) with x < y the sorted non-root vertices, each term (links0 / c0) * (links1 / (c1 + c2))
    from three fresh binomials, zero as soon as an estimate is zero; the expectation of min(1, K / (est p_hat)) over the six
    independent binomials is summed exactly over their (pruned at 1e-13 total mass) product support.
The APX-DD order (:52-168) is deterministic on such graphs: its sampled "fraction of later neighbours" is compared with
2*eta ~ 0.03, and the true fractions are 0 or >= 1/deg; order() restates it with the exact fractions and asserts that margin.
"""
import math

import numpy as np


def cut_params(k, alpha, beta, delta):
    """(h, ell) of EstimateCuts, reference :184-196"""
    ell_raw = 1.0 / (k * delta * alpha * alpha)
    hd = ell_raw * ell_raw * math.log(k / beta)
    if math.isinf(hd) or hd > 100:
        h = 100
    elif hd < 10.0:
        h = 10
    else:
        h = int(math.ceil(hd))
    return h, min(ell_raw, h * 0.5)


def adjacency(n, edges):
    adj = [set() for _ in range(n)]
    for u, v in edges:
        adj[u].add(v)
        adj[v].add(u)
    return [sorted(a) for a in adj]


def order(adj, k, epsilon):
    """(pos, est) of the APX-DD order, reference :52-168, with exact neighbour fractions in place of the sampled ones."""
    n = len(adj)
    beta = epsilon / 2.0
    eta = beta ** (1.0 / (k - 1)) / (6.0 * k * k)
    h = int(math.ceil(10.0 / (eta * eta) * math.log(n)))
    deg = [len(a) for a in adj]
    score = [float(d) for d in deg]
    key = lambda v: (-score[v], -v)
    seq = sorted(range(n), key=key)
    pos = [0] * n
    for i, v in enumerate(seq):
        pos[v] = i
    est = [0.0] * n
    for idx in range(n):
        v = seq[idx]
        d = deg[v]
        if d == 0:
            continue
        frac = sum(1 for u in adj[v] if pos[v] < pos[u]) / d
        sigma = math.sqrt(max(frac * (1 - frac), 1e-12) / h)
        assert abs(frac - 2.0 * eta) > 8 * sigma + 1e-9, "order not deterministic on this graph: sampled fraction too close to its threshold"
        if frac >= 2.0 * eta:
            est[v] = float(d) ** 5
        else:
            score[v] = 3.0 * eta * d
            seq[idx:] = sorted(seq[idx:], key=key)
            for i in range(idx, n):
                pos[seq[i]] = i
    small = k / eta
    for v in range(n):
        if deg[v] > small:
            continue
        queue, seen, enough = [v], {v}, False
        while queue and len(seen) < k:
            u = queue.pop(0)
            for w in adj[u]:
                if w not in seen and pos[v] < pos[w]:
                    seen.add(w)
                    queue.append(w)
                    if len(seen) >= k:
                        enough = True
                        break
        if enough:
            est[v] = float(sum(1 for w in adj[v] if pos[v] < pos[w])) ** 5
        else:
            est[v] = 0.0
    return pos, est


def _binom(h, f):
    j = np.arange(h + 1)
    if f <= 0.0:
        p = np.zeros(h + 1)
        p[0] = 1.0
        return p
    if f >= 1.0:
        p = np.zeros(h + 1)
        p[h] = 1.0
        return p
    logp = np.array([math.lgamma(h + 1) - math.lgamma(i + 1) - math.lgamma(h - i + 1) for i in j]) + j * math.log(f) + (h - j) * math.log(1 - f)
    return np.exp(logp)


def _cut_dist(d, f, h, ell):
    """distribution of one cut estimate: list of (value, probability), zero lumped, negligible mass dropped"""
    p = _binom(h, f)
    zero = float(p[np.arange(h + 1) < ell].sum())
    out = [(0.0, zero)] if zero > 0 else []
    for hits in range(int(math.ceil(ell)), h + 1):
        if p[hits] > 1e-15:
            out.append((d * hits / h, float(p[hits])))
    return out


def law_k3(adj, pos, est, epsilon):
    """{(v, a, b): probability} -- the law of the ORDERED graphlet the reference returns for k = 3."""
    k, C1, C2 = 3, 2, 2
    n = len(adj)
    deg = [len(a) for a in adj]
    beta = epsilon / 2.0
    alpha = beta ** (1.0 / (k - 1)) / (6.0 * k ** 3)
    gamma = epsilon * 3.0 ** (-k) * float(k) ** (-C2)
    rho = gamma
    hg, lg = cut_params(k, alpha, beta, gamma / k ** 4)
    hp, lp = cut_params(k, alpha, beta / k ** 6, rho / (k * k))
    Z = sum(est)
    K = (beta / Z) * float(k) ** (-C1)
    later = lambda v, w: pos[v] < pos[w]
    isadj = lambda u, w: w in adj[u]

    def frac(v, u, U):      # fraction of u's neighbours after v in the order and outside U
        return sum(1 for w in adj[u] if later(v, w) and w not in U) / deg[u] if deg[u] else 0.0

    law = {}
    for v in range(n):
        if not est[v] > 0.0:
            continue
        ok1 = [w for w in adj[v] if later(v, w)]
        if not ok1:
            continue
        q1 = sum(p for c, p in _cut_dist(deg[v], frac(v, v, {v}), hg, lg) if c > 0)
        for a in ok1:
            dv = _cut_dist(deg[v], frac(v, v, {v, a}), hg, lg)
            da = _cut_dist(deg[a], frac(v, a, {v, a}), hg, lg)
            pv = sum(p1 * p2 * c1 / (c1 + c2) for c1, p1 in dv for c2, p2 in da if c1 + c2 > 0)
            pa = sum(p1 * p2 * c2 / (c1 + c2) for c1, p1 in dv for c2, p2 in da if c1 + c2 > 0)
            okv = [w for w in adj[v] if later(v, w) and w not in (v, a)]
            oka = [w for w in adj[a] if later(v, w) and w not in (v, a)]
            for b in sorted(set(okv) | set(oka)):
                pb = (pv / len(okv) if b in okv else 0.0) + (pa / len(oka) if b in oka else 0.0)
                grow = q1 / len(ok1) * pb
                if grow <= 0.0:
                    continue
                # E[accept]: p_hat = p(v,x,y) + p(v,y,x), x < y
                x, y = sorted((a, b))
                terms = []
                for s, t in ((x, y), (y, x)):
                    l0 = 1 if isadj(v, s) else 0
                    l1 = (1 if isadj(v, t) else 0) + (1 if isadj(s, t) else 0)
                    d0 = _cut_dist(deg[v], frac(v, v, {v}), hp, lp)
                    d1 = _cut_dist(deg[v], frac(v, v, {v, s}), hp, lp)
                    d2 = _cut_dist(deg[s], frac(v, s, {v, s}), hp, lp)
                    c0, p0 = (np.array(z) for z in zip(*d0))
                    c1, p1 = (np.array(z) for z in zip(*d1))
                    c2, p2 = (np.array(z) for z in zip(*d2))
                    c12 = c1[:, None] + c2[None, :]
                    # reference :350-360: the product is zeroed by the first zero estimate (the loop over i breaks)
                    with np.errstate(divide="ignore", invalid="ignore"):
                        f0 = np.where(c0 > 0, l0 / np.where(c0 > 0, c0, 1.0), 0.0)
                        f12 = np.where(c12 > 0, l1 / np.where(c12 > 0, c12, 1.0), 0.0)
                    val = (f0[:, None, None] * f12[None, :, :]).ravel()
                    wgt = (p0[:, None, None] * (p1[:, None] * p2[None, :])[None, :, :]).ravel()
                    xs, inv = np.unique(val, return_inverse=True)
                    ps = np.bincount(inv, weights=wgt)
                    keep = ps > 1e-14
                    terms.append((xs[keep], ps[keep]))
                (xa, pa_), (xb, pb_) = terms
                tot = xa[:, None] + xb[None, :]
                w = pa_[:, None] * pb_[None, :]
                with np.errstate(divide="ignore"):
                    acc = np.where(tot > 0, np.minimum(1.0, K / (est[v] * np.where(tot > 0, tot, 1.0))), 0.0)
                e_acc = float((w * acc).sum())
                law[(v, a, b)] = (est[v] / Z) * grow * e_acc
    s = sum(law.values())
    return {key: val / s for key, val in law.items()}, s


def chi_square_p(counts, law):
    """p-value of Pearson's chi-square of observed counts against the law (cells with expectation < 5 are pooled)"""
    from scipy import stats
    n = sum(counts.values())
    assert set(counts) <= set(law), f"samples outside the support of the law: {set(counts) - set(law)}"
    obs, exp, po, pe = [], [], 0.0, 0.0
    for key, p in sorted(law.items()):
        o, e = counts.get(key, 0), n * p
        if e < 5.0:
            po += o
            pe += e
        else:
            obs.append(o)
            exp.append(e)
    if pe > 0:
        obs.append(po)
        exp.append(pe)
    chi2 = sum((o - e) ** 2 / e for o, e in zip(obs, exp))
    return float(stats.chi2.sf(chi2, len(obs) - 1)), chi2, len(obs) - 1


def law_k(adj, pos, est, epsilon, k, mc=200_000, seed=12345):
    """{(v, s1, ..., s_{k-1}): probability}, total acceptance per trial -- the law of the ORDERED graphlet for any k >= 3
    (used for k = 4; law_k3 is the fully enumerated special case and the check of this function).

    Root and growth factors are enumerated exactly like law_k3: at growth step i the i cut estimates are independent binomial
    counts, and `from` is S[j] with probability E[c_j / sum c] (reference :271-300).  The acceptance factor
    E[min(1, K / (est(v) p_hat))] (reference :318-382, :440-447) sums p_hat over the (k-1)! orders of the non-root vertices, each
    order a product of k-1 ratios links_i / (sum of i+1 fresh cut estimates): for k = 4 that is 36 independent binomials per
    graphlet, whose joint support is too large to enumerate -- the expectation is taken over `mc` joint draws of those binomials
    from a fixed generator (relative error of the factor ~ 1/sqrt(mc), far below what a chi-square test on thousands of rows
    resolves).  The factor depends only on (v, set of the others), so it is computed once per vertex set."""
    import itertools
    C1, C2 = 2, 2
    n = len(adj)
    deg = [len(a) for a in adj]
    beta = epsilon / 2.0
    alpha = beta ** (1.0 / (k - 1)) / (6.0 * k ** 3)
    gamma = epsilon * 3.0 ** (-k) * float(k) ** (-C2)
    rho = gamma
    hg, lg = cut_params(k, alpha, beta, gamma / k ** 4)
    hp, lp = cut_params(k, alpha, beta / k ** 6, rho / (k * k))
    Z = sum(est)
    K = (beta / Z) * float(k) ** (-C1)
    later = lambda v, w: pos[v] < pos[w]
    rng = np.random.default_rng(seed)

    def frac(v, u, U):
        return sum(1 for w in adj[u] if later(v, w) and w not in U) / deg[u] if deg[u] else 0.0

    def step_probs(v, S):
        """{w: P(next vertex = w | grown so far = S)} (missing mass = the trial fails here)"""
        dists = [_cut_dist(deg[u], frac(v, u, set(S)), hg, lg) for u in S]
        pick = [0.0] * len(S)
        for combo in itertools.product(*dists):
            tot = sum(c for c, _ in combo)
            if tot <= 0:
                continue
            p = 1.0
            for _, q in combo:
                p *= q
            for j, (c, _) in enumerate(combo):
                pick[j] += p * c / tot
        out = {}
        for j, u in enumerate(S):
            ok = [w for w in adj[u] if later(v, w) and w not in S]
            for w in ok:
                out[w] = out.get(w, 0.0) + pick[j] / len(ok)
        return out

    def draw_cut(u, v, U, size):
        f = frac(v, u, U)
        hits = rng.binomial(hp, f, size=size) if 0.0 < f < 1.0 else np.full(size, hp if f >= 1.0 else 0)
        return np.where(hits >= lp, deg[u] * hits / hp, 0.0)

    acc_cache = {}

    def accept(v, rest):
        key = (v, frozenset(rest))
        if key in acc_cache:
            return acc_cache[key]
        p_hat = np.zeros(mc)
        for perm in itertools.permutations(sorted(rest)):
            seq = (v,) + perm
            p = np.ones(mc)
            for i in range(k - 1):
                Si = seq[: i + 1]
                links = sum(1 for u in Si if seq[i + 1] in adj[u])
                ci = np.zeros(mc)
                for u in Si:
                    ci += draw_cut(u, v, set(Si), mc)
                with np.errstate(divide="ignore", invalid="ignore"):
                    p = np.where((ci > 0) & (p > 0), p * links / np.where(ci > 0, ci, 1.0), 0.0)
            p_hat += p
        with np.errstate(divide="ignore"):
            a = np.where(p_hat > 0, np.minimum(1.0, K / (est[v] * np.where(p_hat > 0, p_hat, 1.0))), 0.0)
        acc_cache[key] = float(a.mean())
        return acc_cache[key]

    law = {}

    def extend(v, S, p):
        if len(S) == k:
            law[tuple(S)] = (est[v] / Z) * p * accept(v, S[1:])
            return
        for w, q in sorted(step_probs(v, S).items()):
            if q > 0.0:
                extend(v, S + [w], p * q)

    for v in range(n):
        if est[v] > 0.0:
            extend(v, [v], 1.0)
    s = sum(law.values())
    return {key: val / s for key, val in law.items()}, s
