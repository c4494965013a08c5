"""Token model of the BGZF sink (sg_deflate.hip) with an optional position-aware source stage for base lines."""
import heapq, sys, collections
import numpy as np

CH = 32768
LANE = 64


def huff_lengths(cnt):
    items = [(c + 1, i) for i, c in enumerate(cnt)]
    heap = [(c, i, None, None) for c, i in items]
    heapq.heapify(heap)
    uid = len(cnt)
    while len(heap) > 1:
        a = heapq.heappop(heap); b = heapq.heappop(heap)
        heapq.heappush(heap, (a[0] + b[0], uid, a, b)); uid += 1
    L = [0] * len(cnt)
    def walk(n, d):
        if n[2] is None: L[n[1]] = max(d, 1)
        else: walk(n[2], d + 1); walk(n[3], d + 1)
    walk(heap[0], 0)
    return L


def len_sym(l):
    if l == 258: return 28, 0
    l3 = l - 3
    if l3 < 8: return l3, 0
    lg = l3.bit_length() - 1
    return 4 * lg - 4 + ((l3 >> (lg - 2)) & 3), lg - 2


def dist_sym(d):
    d1 = d - 1
    if d1 < 4: return d1, 0
    lg = d1.bit_length() - 1
    return 2 * lg + ((d1 >> (lg - 1)) & 1), lg - 1


def acgt(c): return c in b"ACGT"


def tokens(m, positional, stats):
    """m: bytes of a member. returns list of (pos, len, dist) matches sorted."""
    n = len(m)
    # line structure
    lines = []  # (start, end, kind) kind 0 name 1 bases 2 plus 3 qual
    p = 0; k = None
    # find first '@' name line: members start mid-record; classify by walking from a known record start is hard -> use heuristics:
    # a base line = a line made only of ACGTN; quality line follows '+' line.
    starts = [0] + [i + 1 for i in range(n) if m[i] == 10]
    kinds = {}
    ls = [(s, (starts[j + 1] - 1 if j + 1 < len(starts) else n)) for j, s in enumerate(starts)]
    prev_plus = False
    base_lines = []
    for (s, e) in ls:
        seg = m[s:e]
        if len(seg) > 20 and all(c in b"ACGTN" for c in seg) and not prev_plus:
            base_lines.append((s, e))
        prev_plus = (seg == b"+")
    first = {}
    for q in range(0, n - 7, 2):
        g = m[q:q + 8]
        if g not in first: first[g] = q
    matches = []
    cover = bytearray(n)
    # ---- positional stage: for each base line, sources = previous <= 24 base lines in this member
    if positional:
        G = positional
        idx = {}  # gram -> list of (line no, offset) latest last
        for li, (s, e) in enumerate(base_lines):
            line = m[s:e]
            # walk lanes that intersect this line
            p = s
            src = None
            while p < e:
                lane_end = min(e, (p // LANE + 1) * LANE)
                # find source at p: nearest previous line containing gram at p
                best = None
                if src is not None:
                    # continue with the same source if it still matches
                    sp = p - src
                    if sp >= 0 and m[sp:sp + 4] == m[p:p + 4] and sp + 4 <= p:
                        best = sp
                if best is None:
                    g = m[p:p + G]
                    if len(g) == G and g in idx:
                        best = idx[g]
                if best is None:
                    p += 1
                    src = None
                    continue
                # extend to lane end
                l = 0
                while p + l < lane_end and m[best + l] == m[p + l]: l += 1
                if l >= stats["pmin"]:
                    matches.append((p, l, p - best))
                    for t in range(p, p + l): cover[t] = 1
                    src = p - best
                    p += l
                    if p < lane_end:  # mismatch: skip one byte, keep source
                        p += 1
                else:
                    src = None
                    p += 1
            for o in range(0, len(line) - G + 1):
                idx[line[o:o + G]] = s + o   # latest wins
    # ---- hash stage per lane
    for lane in range(n // LANE + (1 if n % LANE else 0)):
        a = lane * LANE; b = min(n, a + LANE)
        lo = a; nm = sum(1 for (p, l, d) in matches if a <= p < b)
        k = a
        while k < b and nm < 6:
            if (k - a) % 4 > 1 or k < lo or k + 8 > b or cover[k]:
                k += 1; continue
            g = m[k:k + 8]
            if k >= 1 and all(m[k - 1 + t] == m[k - 1] for t in range(9)) :
                k += 1; continue
            cq = first.get(g)
            if cq is None or cq >= k:
                k += 1; continue
            l = 8
            while k + l < b and m[cq + l] == m[k + l] and not cover[k + l]: l += 1
            back = 0
            while back < 8 and k - back - 1 >= lo and cq - back - 1 >= 0 and m[k - back - 1] == m[cq - back - 1] and not cover[k - back - 1]: back += 1
            s0 = k - back; l += back
            if l < 12:
                votes = acgt(g[0]) + acgt(g[3]) + acgt(g[7])
                if votes < 3 and 35 not in g:
                    k = max(k + 1, s0 + l - 7)
                    continue
            matches.append((s0, l, k - cq))
            for t in range(s0, s0 + l): cover[t] = 1
            nm += 1
            lo = s0 + l
            k = lo
        # runs
        k = max(a, 1)
        while k < b and nm < 6:
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
    # merge inside groups of 4 lanes
    out = []
    for (p, l, d) in matches:
        if out:
            pp, pl, pd = out[-1]
            if pp + pl == p and pd == d and p % LANE == 0 and (pp // 256) == ((p + l - 1) // 256) and pl + l <= 256:
                out[-1] = (pp, pl + l, d); continue
        out.append((p, l, d))
    return out, cover, base_lines


def run(path, first_member, n_members, positional, pmin=6):
    data = open(path, "rb").read()
    lit = [0] * 286; dist = [0] * 30
    stats = {"pmin": pmin}
    extra = 0
    per = []
    base_cov = base_tot = 0
    for c in range(first_member, first_member + n_members):
        m = data[c * CH:(c + 1) * CH]
        ms, cover, bl = tokens(m, positional, stats)
        for (s, e) in bl:
            base_tot += e - s; base_cov += sum(cover[s:e])
        for i, ch in enumerate(m):
            if not cover[i]: lit[ch] += 1
        lit[256] += 1
        for (p, l, d) in ms:
            s, eb = len_sym(l); lit[257 + s] += 1; extra += eb
            s, eb = dist_sym(d); dist[s] += 1; extra += eb
        per.append(len(ms))
    LL = huff_lengths(lit); DL = huff_lengths(dist)
    bits = sum(c * l for c, l in zip(lit, LL)) + sum(c * l for c, l in zip(dist, DL)) + extra
    hdr = 26 * 8 + 600
    tot = bits + n_members * hdr
    litbits = sum(c * l for c, l in zip(lit[:256], LL[:256]))
    print(f"positional={positional} pmin={pmin}: per member {bits / n_members:.0f} bits; literals {sum(lit[:256]) / n_members:.0f} = {litbits / n_members:.0f} bits; matches {np.mean(per):.0f}; "
          f"ratio {n_members * CH * 8 / tot:.3f}; base bytes covered {base_cov / base_tot:.3f}")
    print("   dist lens", DL)


if __name__ == "__main__":
    path = sys.argv[1]
    nm = int(sys.argv[2])
    run(path, 150, nm, 0)
    for G in (8, 12):
        for pmin in (6, 8):
            run(path, 150, nm, G, pmin)
