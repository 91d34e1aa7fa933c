import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import test_gpu_parity as T
from raytrace_amd import abi, phys, scenes
src = open(T.__file__).read()
# rebuild the scene of the mixed test
import types
ns = {}
body = src[src.index("def test_mixed_scene_random_rays_bit_exact(ob):"):src.index("    with phys.DeviceScene(sc) as dev, ob.OracleScene(sc) as orc:\n        c = orc.intersect(o, d, 1e-6, MAXF)\n        g = dev")]
body = body.replace("def test_mixed_scene_random_rays_bit_exact(ob):", "def build():") + "    return sc, o, d\n"
exec("import numpy as np\nfrom raytrace_amd import abi, phys, scenes\n" + body, ns)
sc, o, d = ns["build"]()
with phys.DeviceScene(sc) as dev:
    g = dev.intersect(o, d, 1e-6, 1.7976931348623157e308, abi.MODE_REFERENCE)
    f = dev.intersect(o, d, 1e-6, 1.7976931348623157e308, abi.MODE_FAST)
bad = np.nonzero(g["node"] != f["node"])[0]
print(len(bad), "mismatches")
import collections
print(collections.Counter(zip(g["node"][bad].tolist(), f["node"][bad].tolist())))
for i in bad[:5]:
    print(i, o[i], d[i], g["node"][i], g["t"][i], f["node"][i], f["t"][i])
