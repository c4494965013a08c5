"""Bits by line type: our token model vs an unrestricted greedy hash-chain parser (zlib-like)."""
import sys, collections
import numpy as np
from tokens import *

def line_types(m):
    n = len(m)
    t = bytearray(n)  # 0 name,1 base,2 plus,3 qual
    starts = [0] + [i + 1 for i in range(n) if m[i] == 10]
    prev_plus = False; prev_kind = None
    for j, s in enumerate(starts):
        e = starts[j + 1] if j + 1 < len(starts) else n
        seg = m[s:e - 1] if e <= n and e > s and m[e - 1] == 10 else m[s:e]
        if seg == b"+" and prev_kind == 1: k = 2
        elif prev_kind == 2: k = 3
        elif len(seg) > 20 and all(c in b"ACGTN" for c in seg): k = 1
        elif seg.startswith(b"@") and prev_kind in (3, None): k = 0
        else:
            # first partial line of the member: guess
            k = 1 if all(c in b"ACGTN" for c in seg) and len(seg) > 0 else (3 if j == 0 else 0)
        for i in range(s, e): t[i] = k
        prev_kind = k
    return t

def greedy(m, chain=32, minlen=4, lazy=True, maxlen=258):
    n = len(m)
    head = collections.defaultdict(list)
    def best_at(p):
        g = m[p:p + minlen]
        bl, bd = 0, 0
        if len(g) < minlen: return 0, 0
        for q in reversed(head[g][-chain:]):
            l = 0
            while p + l < n and l < maxlen and m[q + l] == m[p + l]: l += 1
            if l > bl: bl, bd = l, p - q
        if bl == 3 and bd > 4096: return 0, 0
        return bl, bd
    out = []; cover = bytearray(n)
    p = 0
    def insert(p):
        g = m[p:p + minlen]
        if len(g) == minlen: head[g].append(p)
    while p < n:
        l, d = best_at(p)
        if l >= minlen:
            if lazy and p + 1 < n:
                insert(p)
                l2, d2 = best_at(p + 1)
                if l2 > l:
                    p += 1; continue
                ins_from = p + 1
            else:
                ins_from = p
            out.append((p, l, d))
            for t in range(p, p + l): cover[t] = 1
            for t in range(ins_from, p + l): insert(t)
            p += l
        else:
            insert(p); p += 1
    return out, cover

def evaluate(data, first_member, n_members, fn, label):
    lit = [0] * 286; dist = [0] * 30
    toks = []
    for c in range(first_member, first_member + n_members):
        m = data[c * CH:(c + 1) * CH]
        r = fn(m)
        ms, cover = r[0], r[1]
        toks.append((m, ms, cover))
        for i, ch in enumerate(m):
            if not cover[i]: lit[ch] += 1
        lit[256] += 1
        for (p, l, d) in ms:
            lit[257 + len_sym(l)[0]] += 1; dist[dist_sym(d)[0]] += 1
    LL = huff_lengths(lit); DL = huff_lengths(dist)
    by = np.zeros((4, 3)); cnt = np.zeros((4, 3)); byt = np.zeros((4, 2))
    for (m, ms, cover) in toks:
        t = line_types(m)
        for i, ch in enumerate(m):
            if not cover[i]: by[t[i], 0] += LL[ch]; cnt[t[i], 0] += 1; byt[t[i], 0] += 1
        for (p, l, d) in ms:
            s, eb = len_sym(l); s2, eb2 = dist_sym(d)
            k = t[p]
            by[k, 1] += LL[257 + s] + eb; by[k, 2] += DL[s2] + eb2; cnt[k, 1] += 1; byt[k, 1] += l
    tot = by.sum() / n_members
    print(f"{label}: {tot:.0f} bits/member, ratio {CH * 8 / (tot + 26 * 8 + 600):.3f}")
    for k, name in enumerate(("name", "base", "plus", "qual")):
        print(f"   {name}: lit {cnt[k,0]/n_members:6.0f} = {by[k,0]/n_members:6.0f} b | matches {cnt[k,1]/n_members:5.0f} covering {byt[k,1]/n_members:6.0f} B: len {by[k,1]/n_members:6.0f} b dist {by[k,2]/n_members:6.0f} b | total {by[k].sum()/n_members:6.0f}")

if __name__ == "__main__":
    data = open(sys.argv[1], "rb").read()
    nm = int(sys.argv[2])
    evaluate(data, 150, nm, lambda m: tokens(m, 0, {"pmin": 6})[:2], "ours")
    evaluate(data, 150, nm, lambda m: greedy(m, 32, 4, True), "greedy lazy chain32 min4")
    evaluate(data, 150, nm, lambda m: greedy(m, 32, 4, False), "greedy nolazy chain32 min4")
    evaluate(data, 150, nm, lambda m: greedy(m, 4, 4, False), "greedy nolazy chain4 min4")
    evaluate(data, 150, nm, lambda m: greedy(m, 32, 6, False), "greedy nolazy chain32 min6")
