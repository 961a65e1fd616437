#!/bin/bash
# usage: tools/gr.sh TAG TIMEOUT 'command'   -> runs the command on the GPU box with gpurun_out/r03 present, log in /tmp/gr_TAG.log
TAG=$1; TO=$2; shift 2
/usr/local/graft/bin/gpurun --timeout $TO -- "mkdir -p gpurun_out/r03 && $*" > /tmp/gr_$TAG.log 2>&1
echo "exit $?" >> /tmp/gr_$TAG.log
