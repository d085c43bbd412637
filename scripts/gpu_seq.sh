#!/bin/bash
# Run GPU steps one after the other on the box: `scripts/gpu_seq.sh OUTDIR "name|seconds|command" ...`.  Each step runs under
# `timeout -k 10`, its output goes to OUTDIR/name.log, and the sequence STOPS at the first step that was killed, timed out or
# crashed (exit 124 / 137 / 134 / 139): after such an end no further GPU step is started.  An ordinary failure (a red test,
# exit 1) does not stop the sequence.
out=$1; shift
mkdir -p "$out"
for spec in "$@"; do
    name=${spec%%|*}; rest=${spec#*|}; secs=${rest%%|*}; cmd=${rest#*|}
    echo "[gpu_seq] $name: $cmd" | tee -a "$out/seq.log"
    timeout -k 10 "$secs" bash -c "$cmd" > "$out/$name.log" 2>&1
    rc=$?
    echo "[gpu_seq] $name rc $rc" | tee -a "$out/seq.log"
    case $rc in 124|137|134|139) echo "[gpu_seq] stopping: $name ended abnormally" | tee -a "$out/seq.log"; exit $rc;; esac
done
exit 0
