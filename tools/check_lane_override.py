"""A render with 4096-path queues and 64 lanes per pixel forced (6 of them busy): the launch sizes must follow the wave-rounds, not the paths."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from raytrace_amd import phys, scenes
sc = scenes.cornell(dx=80, dy=72, spp=6)
big = phys.Render(sc)
os.environ["RT_QUEUE_RAYS"] = "4096"; os.environ["RT_LOG2_LANES"] = "6"
small = phys.Render(sc)
assert big.Stats.TotalRays == small.Stats.TotalRays and big.Stats.ShadowRays == small.Stats.ShadowRays
d = np.abs(big.Image.astype(int) - small.Image.astype(int)); assert d.max() <= 1, d.max()
print("lanes override + small queue ok", small.Stats.TraverseLaunches)
