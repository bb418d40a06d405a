#!/bin/bash
# Evidence driver (run HERE, in the build container): every number under profiles/${ROUND}_* (ROUND defaults to r06)
# comes from ONE commit.
#   1. refuses to start when the work tree is dirty (tracked changes or untracked, unignored files);
#   2. stamps HEAD into .evidence_head (git-ignored; it travels with the gpurun snapshot, bench.py copies it into every
#      JSON line as "commit");
#   3. one gpurun call of tools/evidence.sh (suite, bench lines, kernel traces, PMC passes);
#   4. tools/summarise.py turns gpurun_out/ev_$ROUND into profiles/${ROUND}_* -- it checks again that HEAD has not moved and
#      the tree is still clean, and writes profiles/${ROUND}_MANIFEST.json (commit, files, time).
# Usage: tools/collect.sh [gpurun timeout seconds]   (environment of evidence.sh is passed through: WITH_TESTS=1 ...)
# ADDENDUM=1: a partial collection after a change to a few kernels (restrict it with BENCH_WL / GROUP_WL / PMC_WL / SKIP_*):
# the files it produces replace their predecessors and are listed, with THEIR commit, under "addenda" of the manifest.
cd "$(dirname "$0")/.." || exit 1
export ROUND=${ROUND:-r06}
if [ -n "$(git status --porcelain)" ]; then echo "work tree is dirty: commit first (evidence is taken from a commit, not from a state)"; git status --short | head; exit 1; fi
git rev-parse HEAD > .evidence_head
rm -rf gpurun_out/ev_$ROUND
ENVS="ROUND=$ROUND"; for v in WITH_TESTS BENCH_WL GROUP_WL PMC_WL CPU_WL GROUP_CPU_WL SKIP_SWEEP SKIP_SMALL SKIP_GLUE; do [ -z "${!v+x}" ] || ENVS="$ENVS $v='${!v}'"; done
/usr/local/graft/bin/gpurun --timeout ${1:-2400} -- "$ENVS bash tools/evidence.sh"
rc=$?
rm -f .evidence_head
[ $rc -eq 0 ] || { echo "gpurun rc=$rc: nothing summarised"; exit $rc; }
python tools/summarise.py
