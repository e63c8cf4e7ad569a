import json, sys, torch
sys.path.insert(0, "/root/repo")
from tools import benchlib
print(json.dumps(benchlib.bench_dropin_trainer(torch.device("cuda:0")), indent=0))
