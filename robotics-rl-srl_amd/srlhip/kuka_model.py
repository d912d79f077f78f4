"""Bookkeeping constants of the Kuka stepper used by bench.py's roofline line.

FLOPS_PER_ENV_STEP counts the float64 operations of one physics step as
implemented in csrc/kuka_core.hpp (action_repeat = 1), an FMA = 2 flops:
  150 Gauss-Seidel sweeps x (7 arm-motor rows x (6 + 14) + 2 button rows x 8)  = 23 400
  M^-1: 7 unit-torque sweeps x 7 links x ~2 x 26                               =  2 500
  ABA backward (7 x ~330) + forward (7 x ~60)                                  =  2 700
  FK x2, IK (J^T J, LDL^T solve), 7 sincos (~40 each), contacts, rows          =  2 400
"""
FLOPS_PER_ENV_STEP = 3.1e4
