"""TEST INFRASTRUCTURE ONLY -- independent restatement of the mashmap statistics
(src/map/include/map_stats.hpp:56-189, src/map/include/computeMap.hpp:234-293) with numpy
float32/float64 for the reference's float/double mix and scipy.stats for the three GSL
distribution functions (gsl_cdf_binomial_Q = binom.sf, gsl_ran_hypergeometric_pdf(k,n1,n2,t)
= hypergeom.pmf(k, n1+n2, n1, t), gsl_cdf_hypergeometric_P = hypergeom.cdf)."""
import bisect
import math

import numpy as np
from scipy import stats

f32 = np.float32


def j2md(j, k):
    j = f32(j)
    if j == 0:
        return f32(1.0)
    if j == 1:
        return f32(0.0)
    ratio = f32(f32(2) * j) / f32(f32(1) + j)
    return f32(1.0 - math.pow(float(ratio), 1.0 / k))


def md2j(d, k):
    sim = f32(f32(1) - f32(d))
    p = math.pow(float(sim), k)
    return f32(p / (2 - p))


def md_lower_bound(d, s, k, ci):
    q2 = f32((1.0 - float(f32(ci))) / 2)
    j = md2j(d, k)
    x = max(int(math.ceil(s * float(j))), 1)
    while x <= s:
        if stats.binom.sf(x - 1, s, float(j)) < float(q2):
            x -= 1
            break
        x += 1
    return j2md(f32(x) / f32(s), k)


def estimate_minimum_hits(s, k, ident):
    mash = f32(1.0 - float(f32(ident)))
    return int(math.ceil(1.0 * s * float(md2j(mash, k))))


def estimate_minimum_hits_relaxed(s, k, ident, ci):
    first = estimate_minimum_hits(s, k, ident)
    relaxed = first
    for i in range(first, -1, -1):
        d = j2md(f32(1.0 * i / s), k)
        id_upper = f32(1.0 - float(md_lower_bound(d, s, k, ci)))
        if id_upper >= f32(ident):
            relaxed = i
        else:
            break
    return relaxed


def sketch_cutoffs(sketch_size, k, ani_diff, ani_diff_conf):
    delta = f32(ani_diff)
    min_p = f32(f32(1) - f32(ani_diff_conf))
    ss = int(min(sketch_size, 1000.0))
    cut = [1] * (ss + 1)

    def dist_diff(cmax, ci):
        pr = 0.0
        for ymax in range(0, cmax + 1):
            pymax = stats.hypergeom.pmf(ymax, 2 * ss - cmax, ss, cmax)
            if delta == 0:
                yc = float(ymax)
            else:
                yc = math.floor(float(md2j(f32(j2md(f32(ymax / ss), k) + delta), k)) * ss)
            acc = stats.hypergeom.cdf(yc - 1, 2 * ss - ci, ss, ci) if yc - 1 >= 0 else 0.0
            pr += pymax * (1 - acc)
            if pr > float(min_p):
                return True
        return pr > float(min_p)

    for cmax in range(1, ss + 1):
        # upper_bound over ci in [0, ss) with comparator (false < ci) := dist_diff(cmax, ci)
        lo, hi = 0, ss
        while lo < hi:
            mid = (lo + hi) // 2
            if dist_diff(cmax, mid):
                hi = mid
            else:
                lo = mid + 1
        cut[cmax] = lo if lo != 0 else 1
    return cut
