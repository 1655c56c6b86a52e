#!/usr/bin/env python
"""Stand-alone graph-replayed GB/s of the HBM-bound sub-stages (panst3r_amd/stagebench.py):  python tools/hbm_stages.py [v1|v2]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from panst3r_amd import hip
from panst3r_amd.stagebench import standalone_hbm_stages
hip.lib()
res = standalone_hbm_stages(torch.device('cuda:0'), sys.argv[1] if len(sys.argv) > 1 else 'v2')
for k, v in res.items():
    print('%-58s %8.1f us  %7.1f GB/s  (%.2f of 8 TB/s)  %d B per launch' % (k, v['avg_us'], v['GBps'], v['frac_of_8TBps'], v['bytes_per_launch']))
print(json.dumps(res))
