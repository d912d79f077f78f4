"""How often does a row of the Kuka solver change its clamp status (at its lower bound / interior / at its upper bound) from one of the
150 sweeps to the next?  Builds an INSTRUMENTED COPY of oracle/kuka_oracle.c under /tmp (never the shipped oracle), runs random-agent
rollouts of the full model and prints, per contact-free step: sweeps with a flip, flips, histogram, at which sweep the last flip
happened.  Round 4 result (64 envs x 1500 steps): 0.52 sweeps with a flip per step (of 150), 66 % of the steps have none, the last
flip mostly within the first 20 sweeps.  Input to the speculative-linear-sweep analysis in profiles/NOTES.md section G.
Usage (CPU, repo root):  python profiles/probes/kuka_clamp_flip_probe.py"""
import ctypes, os, subprocess, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "robotics-rl-srl_amd"))
import numpy as np

src = open("oracle/kuka_oracle.c").read()
a = "    for (it = 0; it < KM_SOLVER_ITERS; it++) {\n        for (jj = 0; jj < nrows; jj++) {"
assert a in src
src = src.replace(a, """    { static long long n_steps=0, n_flip_sweeps=0, n_flips=0, hist[8]={0}; static int last_hist[160]={0};
      int status[MAX_ROWS]; int kk; for (kk=0;kk<nrows;kk++) status[kk]=-2; int flips_this_step=0, last_it=-1;
    for (it = 0; it < KM_SOLVER_ITERS; it++) {
        int flipped=0;
        for (jj = 0; jj < nrows; jj++) {""")
b = "            if (r->obj >= 0) for (i = 0; i < 3; i++) dvo[r->obj][i] += delta * r->Jo[i] / RB_MASS;\n        }\n    }"
assert b in src
src = src.replace(b, """            if (r->obj >= 0) for (i = 0; i < 3; i++) dvo[r->obj][i] += delta * r->Jo[i] / RB_MASS;
            { int st = r->applied <= r->lo ? -1 : r->applied >= r->hi ? 1 : 0; if (status[k] != -2 && status[k] != st) { flipped=1; n_flips++; } status[k]=st; }
        }
        if (flipped) { flips_this_step++; last_it=it; }
    }
    if (nrows == 15) { n_steps++; n_flip_sweeps += flips_this_step; hist[flips_this_step>7?7:flips_this_step]++; if(last_it>=0) last_hist[last_it]++;
      if (n_steps % 20000 == 0) { fprintf(stderr, "free steps %lld: sweeps with a status flip per step %.3f, flips per step %.3f, histogram of flip-sweeps per step (0..7+) %lld %lld %lld %lld %lld %lld %lld %lld\\n", n_steps, (double)n_flip_sweeps/n_steps, (double)n_flips/n_steps, hist[0],hist[1],hist[2],hist[3],hist[4],hist[5],hist[6],hist[7]);
         fprintf(stderr, "   sweep index of the last flip, first 20:"); for(kk=0;kk<20;kk++) fprintf(stderr," %d", last_hist[kk]); { int ss=0; for(kk=100;kk<150;kk++) ss+=last_hist[kk]; fprintf(stderr," ... sweeps 100-149: %d\\n", ss);} } }
    }""")
src = src.replace("#include <math.h>", "#include <math.h>\n#include <stdio.h>", 1)
os.makedirs("/tmp/orc_probe", exist_ok=True)
open("/tmp/orc_probe/kuka_oracle.c", "w").write(src)
for f in os.listdir("oracle"):
    if f.endswith(".h") or (f.endswith(".c") and f != "kuka_oracle.c"):
        open("/tmp/orc_probe/" + f, "w").write(open("oracle/" + f).read())
subprocess.check_call("cd /tmp/orc_probe && gcc -O2 -std=c99 -fPIC -ffp-contract=off -w -shared -o liboracle_probe.so *.c -lm", shell=True)
import oracle.clib as C
C.lib()
C._lib = ctypes.CDLL("/tmp/orc_probe/liboracle_probe.so")
from oracle import kuka_clib
kuka_clib.set_full(True)
out = kuka_clib.rollout(np.arange(64), 1500, actions=None, rng_mode=kuka_clib.RNG_PHILOX, trace=False)
print("episodes finished:", int(out["done"].sum()))
