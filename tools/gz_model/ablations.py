"""Ablations of the lane token model."""
import sys, collections
import numpy as np
from tokens import len_sym, dist_sym, huff_lengths, acgt, CH, LANE
from bits_by_line import line_types, evaluate, greedy

def tokens3(m, probe_res=(0, 1), table_even=True, lane_tail=False, source="first", gram=8, min_copy=12, min_other=8, maxm=6, back_max=8,
            tail_min=5, fixed_dist=False, parts=1):
    n = len(m)
    first = {}
    step = 2 if table_even else 1
    psize = (n + parts - 1) // parts
    for q in range(0, n - gram + 1, step):
        g = m[q:q + gram]
        if source == "first":
            if parts > 1:   # first occurrence inside every part of the member: up to `parts` candidates, the longest wins
                lst = first.setdefault(g, [])
                if not lst or lst[-1] // psize != q // psize: lst.append(q)
            elif g not in first: first[g] = [q]
        else:
            first.setdefault(g, []).append(q)
    matches = []; cover = bytearray(n)
    import bisect
    for lane in range((n + LANE - 1) // LANE):
        a = lane * LANE; b = min(n, a + LANE)
        lo = a; nm = 0; k = a
        while k < b and nm < maxm:
            if (k - a) % 4 not in probe_res or k < lo or (k + gram > b and not lane_tail) or k + gram > n:
                k += 1; continue
            g = m[k:k + gram]
            if k >= 1 and all(m[k - 1 + t] == m[k - 1] for t in range(9) if k - 1 + t < n):
                k += 1; continue
            lst = first.get(g)
            cq = None
            if lst:
                if source == "first" and parts > 1:
                    bestl = -1
                    for cand in lst:
                        if cand >= k: break
                        l_ = 0
                        while k + l_ < b and m[cand + l_] == m[k + l_]: l_ += 1
                        bk = 0
                        while bk < back_max and k - bk - 1 >= lo and cand - bk - 1 >= 0 and m[k - bk - 1] == m[cand - bk - 1]: bk += 1
                        if l_ + bk >= bestl: bestl = l_ + bk; cq = cand
                elif source == "first":
                    cq = lst[0] if lst[0] < k else None
                else:
                    i = bisect.bisect_left(lst, k)
                    cq = lst[i - 1] if i > 0 else None
            if cq is None:
                k += 1; continue
            l = 0
            while k + l < b and m[cq + l] == m[k + l]: l += 1
            back = 0
            while back < back_max and k - back - 1 >= lo and cq - back - 1 >= 0 and m[k - back - 1] == m[cq - back - 1]: back += 1
            s0 = k - back; l += back
            if l < min_copy:
                votes = acgt(g[0]) + acgt(g[3]) + acgt(g[-1])
                if (votes < 3 and 35 not in g) or l < (tail_min if k + gram > b else min_other):
                    k = max(k + 1, s0 + l - 7)
                    continue
            matches.append((s0, l, k - cq))
            for t in range(s0, s0 + l): cover[t] = 1
            nm += 1; lo = s0 + l; k = lo
        k = max(a, 1)
        while k < b and nm < maxm:
            if cover[k] or k + 3 >= n or not (m[k - 1] == m[k] == m[k + 1] == m[k + 2] == m[k + 3]):
                k += 1; continue
            l = 0
            while k + l < b and not cover[k + l] and m[k + l] == m[k - 1]: l += 1
            if l >= 5:
                matches.append((k, l, 1))
                for t in range(k, k + l): cover[t] = 1
                nm += 1
            k += max(l, 1)
    matches.sort()
    out = []
    for (p, l, d) in matches:
        if out:
            pp, pl, pd = out[-1]
            if pp + pl == p and pd == d and p % LANE == 0 and (pp // 256) == ((p + l - 1) // 256) and pl + l <= 256:
                out[-1] = (pp, pl + l, d); continue
        out.append((p, l, d))
    return out, cover

if __name__ == "__main__":
    data = open(sys.argv[1], "rb").read()
    nm = int(sys.argv[2])
    V = {
        "base": {},
        "lane_tail": dict(lane_tail=True),
        "all probes": dict(probe_res=(0, 1, 2, 3), table_even=False),
        "last source": dict(source="last"),
        "last source + all probes": dict(source="last", probe_res=(0, 1, 2, 3), table_even=False),
        "gram6": dict(gram=6, min_other=6),
        "gram6 lane_tail last": dict(gram=6, min_other=6, lane_tail=True, source="last"),
        "maxm 10": dict(maxm=10),
        "parts2": dict(parts=2),
        "parts4": dict(parts=4),
        "parts8": dict(parts=8),
        "parts4 + all probes": dict(parts=4, probe_res=(0, 1, 2, 3), table_even=False),
        "parts16 + all probes": dict(parts=16, probe_res=(0, 1, 2, 3), table_even=False),
        "back 64": dict(back_max=64),
    }
    for k, v in V.items():
        if len(sys.argv) > 3 and sys.argv[3] not in k: continue
        evaluate(data, 150, nm, lambda m: tokens3(m, **v), k)
