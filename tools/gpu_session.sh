#!/bin/bash
# One gpurun call: the GPU test-suite, then the follow-up commands given as arguments — but only if the tests ended in an ordinary
# way (all passed, or assertion failures): after a crash, a hang or a kill nothing else is started on the box.
#   tools/gpu_session.sh <pytest-args...> -- <command> [&& ...]   (the part after -- is run with bash -c)
mkdir -p gpurun_out
args=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do args+=("$1"); shift; done; shift
timeout -k 10 850 python -m pytest "${args[@]}" > gpurun_out/pytest.log 2>&1
rc=$?
tail -n 15 gpurun_out/pytest.log
echo "[gpu_session] pytest rc=$rc"
if [ $rc -ne 0 ] && [ $rc -ne 1 ]; then echo "[gpu_session] abnormal end of the test run: stopping here"; exit $rc; fi
if [ $# -gt 0 ]; then bash -o pipefail -c "$*"; rc2=$?; echo "[gpu_session] follow-up rc=$rc2"; fi
exit $rc
