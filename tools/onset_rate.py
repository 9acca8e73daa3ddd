#!/usr/bin/env python3
"""CPU (fp32 oracle): how often is a headline env COUPLED (a contact between the arm / the drawer and the block / a scene-joint body: the rows that send a
k_solve2 block down its slow two-env path), and how often does that start - bench.py's action distribution, 48 envs x 50 counted steps.
DESIGN.md section 4 quotes it: coupling is transient (an episode lasts ~1.3 steps), so no static "heavy group" can hold the coupled envs.
    python tools/onset_rate.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'oracle'))
from oracle import OracleEnv
LO=np.array([-0.18,0.0,0.05,-0.5,-0.5,-0.5,-1.0]); HI=np.array([0.18,0.3,0.3,0.5,0.5,0.5,1.0])
rng=np.random.default_rng(0)
n,steps=48,60
o=[OracleEnv('U',seed=1234,env_index=e,f32=True) for e in range(n)]
cols=o[0].collider_list(); body=[c['body'] for c in cols]
def coupled(env):
    k=0
    for c in env.contacts():
        a,b=body[int(c[0])],body[int(c[1])]
        r0=lambda x: (1<=x<=12) or x==14
        r1=lambda x: x==13 or x>=15
        if (r0(a) and r1(b)) or (r0(b) and r1(a)): k+=1
    return k
for e in o: e.reset()
prev=np.zeros(n,bool); tot=0; on=0; cnt=0
for t in range(steps):
    for i,e in enumerate(o):
        e.step(LO+(HI-LO)*rng.random(7))
        c=coupled(e)>0
        if t>=10:
            tot+=1; cnt+=c; on+= (c and not prev[i])
        prev[i]=c
print('env-steps',tot,'coupled at step end %.2f%%'%(100*cnt/tot),'onsets (coupled now, not at previous step end) %.2f%% of env-steps'%(100*on/tot))
