import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), 'robotics-rl-srl_amd'))
import numpy as np
from oracle import kuka_clib
from srlhip import _lib
n=8
for kern in ("lane","group"):
    os.environ["SRLHIP_KUKA_KERNEL"]=kern
    cfg=_lib.default_config(_lib.ENV_KUKA_BUTTON); cfg.num_envs=n; cfg.seed0=3; cfg.rng_mode=_lib.RNG_MT19937; cfg.auto_reset=1
    h=_lib.Handle(cfg); h.reset()
    a=np.full(n,4,np.int32)
    o,r,d=h.step(a)
    print(kern,'rew',r[:4],'done',d[:4],'obs',o[0])
    print('  q',h.get_state(_lib.F_KUKA_Q).T[0]); print('  qd',h.get_state(_lib.F_KUKA_QD).T[0])
    print('  grip',h.get_state(_lib.F_KUKA_GRIPPER).T[0],'bq',h.get_state(_lib.F_KUKA_BUTTON_Q).T[0],'cnt',h.get_state(_lib.F_KUKA_COUNTERS).T[0], 'ee', h.get_state(_lib.F_KUKA_EE_TARGET).T[0])
    h.close()
