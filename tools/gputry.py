import numpy as np, sys, time
sys.path.insert(0, ".")
from microservice_matchmaking_amd import Engine, make_config, mode_1v1
from microservice_matchmaking_amd.synth import make_pool
cfg = make_config([mode_1v1(window=25, region_filter=True)], capacity=1<<17)
r, c = make_pool(int(sys.argv[1]), seed=3)
with Engine(cfg) as e:
    e.enqueue(r, c)
    t0=time.time(); m = e.tick(0)
    print("ok", sys.argv[1], len(m), m.stats["passes_max"], "%.1f ms" % ((time.time()-t0)*1e3), flush=True)
