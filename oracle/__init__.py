"""CPU oracle — TEST INFRASTRUCTURE ONLY.

Restatement of the env-step hot path of araffin/robotics-rl-srl (reference
checked out at /root/reference when the fixtures were generated).  Only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this package; the product (``robotics-rl-srl_amd/``) never
does and fails loudly when the HIP library is missing.

Pinning status:
  * MobileRobot family  — PINNED against golden vectors produced by the
    reference's own Python source (tests/golden/make_mobile_golden.py imports
    /root/reference/environments/mobile_robot/*.py with pybullet/gym stubbed).
  * Kuka control / reward / termination wrapper — PINNED the same way against
    kuka.py + kuka_button_gym_env.py driven by a scripted fake ``pybullet``.
  * Kuka rigid-body dynamics (pybullet==1.8.6, absent) — PARITY UNPINNED:
    restated from Bullet's published algorithm (Featherstone ABA +
    projected Gauss-Seidel), cross-checked only against an independent
    numpy CRBA/RNEA formulation.
"""
