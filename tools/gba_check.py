import importlib, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from oracle_api import Oracle, build_oracle
build_oracle(); o = Oracle()
pkg = importlib.import_module("orb_slam3-1_amd"); synth = importlib.import_module("orb_slam3-1_amd.synth")
for n_opt in (50, 100, 150, 190):
    w = synth.make_ba_window(3, n_opt=n_opt, n_fixed=10, n_points=8000 * n_opt // 190, obs_per_point=10)
    r0 = o.lba_solve(w, 5)
    s = pkg.LbaSolver()
    t = time.time(); r1 = s.solve(w, 5); dt = time.time() - t
    s.close()
    print(n_opt, "oracle", r0["stats"]["iterations"], r0["stats"]["trials"], "%.6g" % r0["stats"]["chi2_final"],
          "| hip", r1["stats"]["iterations"], r1["stats"]["trials"], "%.6g" % r1["stats"]["chi2_final"], "stop", r1["stats"]["stop_reason"], "%.1f ms" % (1e3 * dt))
