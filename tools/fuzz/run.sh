#!/bin/bash
# AddressSanitizer + UBSan runs of the host code of the BAM side (no GPU): the one-shot inflater on valid, truncated and
# mutated deflate streams with exact-size buffers, the Huffman BGZF encoder against zlib's inflate, the CIGAR walk of
# rmr_ref_to_signal on random and hostile CIGARs, the batch forms of round 5 (rmr_ref_anchor_batch, rmr_orient_bases) and the
# set-order restatement of csrc/pyset_order.c.  From the repository root:
#   bash tools/fuzz/run.sh
set -e
cd "$(dirname "$0")"
g++ -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=undefined -o /tmp/rmr_inflate_asan inflate_asan.cpp -lz
g++ -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=undefined -pthread -I../../include -o /tmp/rmr_bgzf_asan bgzf_asan.cpp ../../remora_amd/csrc/bgzf_deflate.cpp -lz
g++ -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=undefined -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -o /tmp/rmr_r2s_asan r2s_asan.cpp
g++ -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=undefined -pthread -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -I../../include -o /tmp/rmr_batch_asan batch_asan.cpp
gcc -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=undefined -pthread -o /tmp/rmr_setorder_asan setorder_asan.c ../../remora_amd/csrc/pyset_order.c
/tmp/rmr_inflate_asan
/tmp/rmr_bgzf_asan
/tmp/rmr_r2s_asan
/tmp/rmr_batch_asan
/tmp/rmr_setorder_asan
