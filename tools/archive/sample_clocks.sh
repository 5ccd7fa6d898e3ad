#!/bin/bash
# Shader clock / power while the evaluation loop runs (DVFS check): samples rocm-smi every 0.5 s next to run_scene.py.
REPO=${GRAFT_REPO_ROOT:-/root/repo}
[ -f /tmp/scene.npz ] || python $REPO/tools/make_scene_cache.py /tmp/scene.npz > /tmp/make_scene.log 2>&1
python $REPO/tools/run_scene.py /tmp/scene.npz ${1:-40000} fp64 256 0 0 > /tmp/clk_run.log 2>&1 &
PID=$!
for i in $(seq 1 24); do
  sleep 0.5
  kill -0 $PID 2>/dev/null || break
  /opt/rocm/bin/rocm-smi --showclocks --showpower 2>/dev/null | grep -i -E "sclk|mclk|power" | tr -s ' ' | tr '\n' '|'
  echo
done
wait $PID
tail -c 300 /tmp/clk_run.log
