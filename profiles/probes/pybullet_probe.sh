#!/bin/bash
# Round-5 verdict item 2: is a PyBullet wheel (the reference's L0: kuka.py:60, kuka_button_gym_env.py:219-223) reachable from the GPU box?
# Writes everything it finds to gpurun_out/r05_pybullet_probe.log.  Bounded: every network call has a short timeout.
mkdir -p gpurun_out
L=gpurun_out/r05_pybullet_probe.log
{
echo "== date: $(date -u)"; echo "== host: $(hostname)"
echo "== python -c 'import pybullet'"; python -c 'import pybullet, pybullet_data; print(pybullet.__file__, pybullet_data.getDataPath())' 2>&1 | tail -2
echo "== pip config / index"; pip config list 2>&1; env | grep -i -E 'pip_|proxy|index' 2>&1
echo "== find wheels / data on disk"
find / -xdev \( -iname 'pybullet*' -o -name 'kuka_with_gripper2.sdf' -o -name 'kuka_iiwa' -o -iname 'bullet3*' -o -name 'libBullet*' \) 2>/dev/null | grep -v '^/proc' | head -40
echo "== wheelhouse dirs"; find / -xdev -type d \( -iname '*wheelhouse*' -o -iname 'wheels' \) 2>/dev/null | head
echo "== DNS / TCP reachability"
for h in pypi.org files.pythonhosted.org github.com; do
  timeout 8 getent hosts $h && echo "dns $h ok" || echo "dns $h FAIL"
  timeout 8 bash -c "exec 3<>/dev/tcp/$h/443" 2>&1 && echo "tcp $h:443 ok" || echo "tcp $h:443 FAIL"
done
echo "== pip download pybullet==1.8.6"; (cd /tmp && timeout 60 pip download --no-deps --timeout 8 --retries 0 pybullet==1.8.6 2>&1 | tail -5)
echo "== pip download pybullet (any)"; (cd /tmp && timeout 60 pip download --no-deps --timeout 8 --retries 0 pybullet 2>&1 | tail -5)
echo "== pip install --no-index pybullet (offline wheelhouse?)"; (timeout 60 pip install --no-index pybullet 2>&1 | tail -3)
echo "== done"
} > $L 2>&1
cat $L
