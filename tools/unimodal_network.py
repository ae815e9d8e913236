#!/usr/bin/env python3
"""Developer tool (CPU): minimum-size comparator networks that sort every unimodal (ascending, then descending) sequence
of n keys -- the shape min(A[i], B[n-1-i]) of two ascending lists has (raster_forward.hip, merge_round / sort_unimodal).
Breadth-first search over the reachable sets of 0/1 inputs 0^a 1^b 0^c (the 0-1 principle holds on a class of inputs that
is closed under monotone maps), then a numeric check against sorted(a + b)[:n] on random lists with ties.
    python tools/unimodal_network.py [n ...]        (n <= 6 finishes in seconds)"""
import random
import sys
from collections import deque


def search(n, max_size=8):
    inputs = frozenset(tuple([0] * a + [1] * b + [0] * (n - a - b)) for a in range(n + 1) for b in range(n + 1 - a))
    comps = [(i, j) for i in range(n) for j in range(i + 1, n)]

    def apply(states, i, j):
        return frozenset(tuple(sorted((s[i], s[j]))[k == j] if k in (i, j) else s[k] for k in range(n)) for s in states)

    seen, queue = {inputs: []}, deque([inputs])
    while queue:
        st = queue.popleft()
        path = seen[st]
        if all(list(s) == sorted(s) for s in st):
            return path
        if len(path) < max_size:
            for c in comps:
                ns = apply(st, *c)
                if ns not in seen:
                    seen[ns] = path + [c]
                    queue.append(ns)
    return None


def check(n, net, trials=100000):
    for _ in range(trials):
        a, b = sorted(random.choices(range(12), k=n)), sorted(random.choices(range(12), k=n))
        s = [min(a[k], b[n - 1 - k]) for k in range(n)]
        for i, j in net:
            if s[i] > s[j]:
                s[i], s[j] = s[j], s[i]
        assert s == sorted(a + b)[:n], (a, b, s)


if __name__ == "__main__":
    for n in [int(x) for x in sys.argv[1:]] or [2, 3, 4, 5, 6]:
        net = search(n)
        check(n, net)
        print(n, len(net), net)
