#!/bin/bash
# one PMC pass: tools/pmc_one.sh <tag> <kernel-pattern> "<counters>" -- <python script + args>
: ${GRAFT_REPO_ROOT:?run through gpurun (or export GRAFT_REPO_ROOT=<repo root>)}
tag=$1; pat=$2; set=$3; shift 4
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pmc_$tag
rocprofv3 --kernel-trace --pmc $set --output-format csv -d gpurun_out/pmc_$tag/p -- python3 "$@" > gpurun_out/pmc_$tag/log.txt 2>&1
python3 tools/pmc_summary.py gpurun_out/pmc_$tag $pat
