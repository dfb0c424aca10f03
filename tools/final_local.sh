#!/bin/bash
# Local front end of the FINAL measurement battery (two rounds running a file called FINAL predated the
# last kernel commit).  Refuses to run unless the work tree is clean, builds the library from the committed sources,
# records "<commit> <kernel-source hash>" in tools/.final_stamp (git-ignored; it travels with the gpurun snapshot) and only
# then calls tools/final_measure.sh on the GPU box, which re-hashes the sources it finds and refuses a mismatch.  Every
# artefact of the battery therefore carries ONE hash = the committed tree's.
#   bash tools/final_local.sh [tag] [gpurun timeout s]
set -e
cd "$(dirname "$0")/.."
if [ -n "$(git status --porcelain --untracked-files=no)" ]; then
  echo "final_local: the work tree has uncommitted changes -- commit first:" >&2
  git status --short --untracked-files=no >&2
  exit 2
fi
make -C maskdit_amd/csrc -j8 > /tmp/mdt_make.log 2>&1 || { grep -i error /tmp/mdt_make.log | head; echo "final_local: build failed" >&2; exit 1; }
HASH=$(python -c "from maskdit_amd import _lib; print(_lib.source_hash())")
echo "$(git rev-parse --short=12 HEAD) $HASH" > tools/.final_stamp
echo "final_local: commit $(git rev-parse --short=12 HEAD), kernel-source hash $HASH"
exec /usr/local/graft/bin/gpurun --timeout "${2:-5400}" -- "bash tools/final_measure.sh ${1:-r6final}"
