"""Prototype: compare BVH builders by SAH cost on the repo's meshes (LBVH as on the GPU, PLOC, binned SAH)."""
import sys; sys.path.insert(0, '/root/repo')
import numpy as np
from raytracer_amd import scenes, host

CI, CT = 1.0, 1.3

def area(lo, hi):
    e = np.maximum(hi - lo, 0); return 2 * (e[..., 0] * e[..., 1] + e[..., 1] * e[..., 2] + e[..., 2] * e[..., 0])

def tri_boxes(geos):
    lo, hi = [], []
    for g in geos:
        p = g['vertices']['position'][g['indices'].astype(int)].reshape(-1, 3, 3)
        lo.append(p.min(1)); hi.append(p.max(1))
    return np.concatenate(lo), np.concatenate(hi)

def morton(lo, hi):
    c = 0.5 * (lo + hi); slo, shi = lo.min(0), hi.max(0)
    q = np.clip(((c - slo) / np.maximum(shi - slo, 1e-30) * 1024).astype(int), 0, 1023)
    def ex(v):
        v = (v * 0x00010001) & 0xFF0000FF; v = (v * 0x00000101) & 0x0F00F00F; v = (v * 0x00000011) & 0xC30C30C3; v = (v * 0x00000005) & 0x49249249; return v
    return (ex(q[:, 0]) << 2) | (ex(q[:, 1]) << 1) | ex(q[:, 2])

class Node:
    __slots__ = ('lo', 'hi', 'l', 'r', 'n', 'cost', 'leaf')

def sah_collapse(node, leaf_max):
    """bottom-up: returns un-normalised cost; marks node.leaf"""
    a = area(node.lo, node.hi)
    if node.l is None:
        node.leaf = True; node.cost = CT * a; return node.cost
    cl, cr = sah_collapse(node.l, leaf_max), sah_collapse(node.r, leaf_max)
    split = CI * a + cl + cr
    leafc = CT * node.n * a
    if node.n <= leaf_max and leafc <= split:
        node.leaf = True; node.cost = leafc
    else:
        node.leaf = False; node.cost = split
    return node.cost

def count_collapse(node, leaf_max):
    a = area(node.lo, node.hi)
    if node.n <= leaf_max:
        node.leaf = True; node.cost = CT * node.n * a; return node.cost
    node.leaf = False
    node.cost = CI * a + count_collapse(node.l, leaf_max) + count_collapse(node.r, leaf_max)
    return node.cost

def mk_leaf(lo, hi):
    n = Node(); n.lo, n.hi, n.l, n.r, n.n = lo, hi, None, None, 1; return n
def mk_int(a, b):
    n = Node(); n.lo, n.hi, n.l, n.r, n.n = np.minimum(a.lo, b.lo), np.maximum(a.hi, b.hi), a, b, a.n + b.n; return n

def lbvh(lo, hi):
    order = np.argsort(morton(lo, hi), kind='stable'); keys = morton(lo, hi)[order]
    def build(f, l):
        if f == l: return mk_leaf(lo[order[f]], hi[order[f]])
        kf, kl = int(keys[f]), int(keys[l])
        if kf == kl: split = (f + l) // 2
        else:
            # highest differing bit
            pref = (kf ^ kl).bit_length()
            split = f
            step = l - f
            while step > 1:
                step = (step + 1) // 2
                ns = split + step
                if ns < l and ((kf ^ int(keys[ns])).bit_length() < pref): split = ns
        return mk_int(build(f, split), build(split + 1, l))
    return build(0, len(order) - 1)

def ploc(lo, hi, r=8):
    order = np.argsort(morton(lo, hi), kind='stable')
    cl = [mk_leaf(lo[i], hi[i]) for i in order]
    it = 0
    while len(cl) > 1:
        n = len(cl); it += 1
        L = np.array([c.lo for c in cl]); H = np.array([c.hi for c in cl])
        nn = np.zeros(n, int)
        for i in range(n):
            j0, j1 = max(0, i - r), min(n, i + r + 1)
            a = area(np.minimum(L[i], L[j0:j1]), np.maximum(H[i], H[j0:j1]))
            a[i - j0] = np.inf
            nn[i] = j0 + int(np.argmin(a))
        out = []
        for i in range(n):
            j = nn[i]
            if nn[j] == i:
                if i < j: out.append(mk_int(cl[i], cl[j]))
            else: out.append(cl[i])
        cl = out
    return cl[0], it

def binned_sah(lo, hi):
    c = 0.5 * (lo + hi)
    def build(ids):
        if len(ids) == 1: return mk_leaf(lo[ids[0]], hi[ids[0]])
        best = None
        for ax in range(3):
            o = ids[np.argsort(c[ids, ax], kind='stable')]
            Ll = np.minimum.accumulate(lo[o], 0); Lh = np.maximum.accumulate(hi[o], 0)
            Rl = np.minimum.accumulate(lo[o][::-1], 0)[::-1]; Rh = np.maximum.accumulate(hi[o][::-1], 0)[::-1]
            k = np.arange(1, len(o))
            cost = area(Ll[:-1], Lh[:-1]) * k + area(Rl[1:], Rh[1:]) * (len(o) - k)
            i = int(np.argmin(cost))
            if best is None or cost[i] < best[0]: best = (cost[i], o, i + 1)
        _, o, s = best
        return mk_int(build(o[:s]), build(o[s:]))
    return build(np.arange(len(lo)))

def stats(root, collapse, leaf_max):
    cost = collapse(root, leaf_max) / area(root.lo, root.hi)
    # expected nodes visited / tris tested for random rays hitting the root box ~ area ratios
    en, et = 0.0, 0.0
    st = [root]; A = area(root.lo, root.hi)
    while st:
        n = st.pop()
        if n.leaf: et += area(n.lo, n.hi) / A * n.n
        else: en += area(n.lo, n.hi) / A; st += [n.l, n.r]
    return cost, en, et

if __name__ == '__main__':
    for name, sc in [('bunny', scenes.bunny(64, 36).blas[0]), ('cornell', scenes.cornell(64, 36).blas[0])]:
        lo, hi = tri_boxes(sc)
        print(name, len(lo), 'tris')
        t = lbvh(lo, hi); print('  lbvh  count<=4 : cost %.2f nodes %.2f tris %.2f' % stats(t, count_collapse, 4))
        print('  lbvh  sah<=4   : cost %.2f nodes %.2f tris %.2f' % stats(t, sah_collapse, 4))
        for r in (4, 8, 16):
            t, it = ploc(lo, hi, r); print('  ploc r=%d (%d it) sah<=4: cost %.2f nodes %.2f tris %.2f' % ((r, it) + stats(t, sah_collapse, 4)))
        t = binned_sah(lo, hi); print('  sweep-sah sah<=4: cost %.2f nodes %.2f tris %.2f' % stats(t, sah_collapse, 4))
