import sys, copy
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np
from emu_fuzz import random_case
from raytracer_amd import _abi
import oracle, raytracer_amd as rt
oracle.build()
seed, idx = int(sys.argv[1]), int(sys.argv[2])
data, cam, params = random_case(seed, idx)
o_rgba, o_hit, o_cnt = oracle.render(data, cam, params, threads=8)
sc = rt.Scene(data)
p = copy.copy(params); p.search = _abi.RTMI_SEARCH_LIST_SCAN
g_rgba, g_hit, g_cnt = sc.render(cam, p)
bad = np.argwhere(g_hit != o_hit)
print("objects", [(int(o['kind']), int(o['first']), int(o['count'])) for o in data.objects])
print("n bad", len(bad), "shape", g_hit.shape)
from collections import Counter
print("expected->got", Counter((int(o_hit[tuple(b)]), int(g_hit[tuple(b)])) for b in bad).most_common(30))
print("bad sample sub-index histogram", Counter(int(b[2]) for b in bad))
print({k: (g_cnt[k], o_cnt[k]) for k in ("samples","primary_hits","tri_tests","tri_tests_primary_kernel","tri_search_launches") if k in g_cnt})
