#!/bin/bash
# Prepare a second copy of the tree at a git ref under .ab/<name> (built in place) so that one gpurun call can time two
# versions of the package on the SAME box: tests/probes/ab_prepare.sh <git-ref> <name>
set -e
cd /root/repo
ref=$1; name=$2
rm -rf .ab/$name; mkdir -p .ab/$name
git archive $ref commu-code_amd include tests/probes bench.py oracle | tar -x -C .ab/$name
python .ab/$name/commu-code_amd/build.py > /dev/null
echo "prepared .ab/$name from $ref"
