"""The captured default-config training step under rocprofv3 --kernel-trace --stats: which kernels a step is made of."""
import sys
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cnf_ot_amd import solvers
kind = sys.argv[1] if len(sys.argv) > 1 else "ot"       # "rwpo": the checked-in default of config/mfc.yaml
config = solvers.load_config(overrides={"general": {"type": kind, "t_batch_size": 1}})
m = solvers.build_model(config); p = m.init(1); opt = solvers.Adam(1e-3); st = opt.init(p)
upd = solvers.CapturedUpdate(solvers.bind_loss(config, m), opt, 2048)
for i in range(203): upd(p, i + 1, 5000.0, st)
torch.cuda.synchronize()
