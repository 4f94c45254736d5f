import json, sys, os
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np
from path_planner_amd import api, _abi as A
ctx=api.Context(0)
legs=json.load(open('/root/repo/tests/golden/rs_demo_legs.json'))
for leg in legs:
    npaths,best,words,flags=api.rs_batch(ctx,[leg['q0']],[leg['q1']],leg['maxc'])
    got=api.calc_all_paths(ctx,*leg['q0'],*leg['q1'],leg['maxc'],leg['step'])
    print(flags[0], npaths[0], best[0], leg['best'], [p['ctypes'] for p in got], [p['ctypes'] for p in leg['paths']], words[0].word(), words[0].total_length, leg['paths'][leg['best']]['total'])
