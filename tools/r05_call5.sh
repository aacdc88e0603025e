#!/bin/bash
# Round-5 fifth GPU call: reads pipeline with the event hand-off / device chunk_read / three workers; the build without
# SLP vectorisation (no packed fp32 op_sel on LDS-fed registers, tools/lint_pk_lds.py) against the shipped one; ingest profile.
set -u
export TMPDIR=/tmp
O=gpurun_out/r05; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fused.py tests/test_gpu_refine.py tests/test_gpu_ingest.py -m gpu -q -k "call_read or call_reads or extract or specified or batched or streamed or subbatch or ingest or real_read or infer or pipelined or prepare" > $O/pytest_call5.txt 2>&1; echo "rc=$?" >> $O/pytest_call5.txt; tail -5 $O/pytest_call5.txt | cut -c1-300
timeout 300 python tools/timeline_reads.py --out $O/timeline_host_call5.md > $O/timeline_host_call5.log 2>&1; tail -7 $O/timeline_host_call5.log
for DT in bf16 fp32 f16x3; do timeout 200 python tools/ab_variants.py --libs default,noslp --dtype $DT > $O/ab_noslp_$DT.log 2>&1; cat $O/ab_noslp_$DT.log | cut -c1-260; done
timeout 200 python tools/ab_variants.py --libs default,noslp --dtype fp32 --arch conv_only > $O/ab_noslp_conv.log 2>&1; cat $O/ab_noslp_conv.log | cut -c1-260
timeout 200 python tools/prof_ingest_batches.py 6000 > $O/prof_ingest_batches.log 2>&1; tail -45 $O/prof_ingest_batches.log | cut -c1-200
