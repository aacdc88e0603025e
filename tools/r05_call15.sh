#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r05; mkdir -p $O
timeout 300 python tools/prof_ingest_batches.py 4000 ref 256 > $O/prof_ingest_ref_b256_sections.log 2>&1; grep -v amdgpu $O/prof_ingest_ref_b256_sections.log | head -4
timeout 300 python tools/prof_ingest_batches.py 4000 ref 512 > $O/prof_ingest_ref_b512_sections.log 2>&1; grep -v amdgpu $O/prof_ingest_ref_b512_sections.log | head -4
timeout 300 python tools/prof_ingest_batches.py 6000 can 512 > $O/prof_ingest_can_b512_sections.log 2>&1; grep -v amdgpu $O/prof_ingest_can_b512_sections.log | head -4
