# which pipelines does a foreign process that keeps the bf16 matrix cores busy disturb?  (profiles/NOTES_r04.md)
#   bash tools/ubench/neighbour_matrix.sh [kind=mfma16] [reps=300] "<mix entries separated by spaces>"
mkdir -p gpurun_out
KIND=${1:-mfma16}; REPS=${2:-300}; shift 2
N=tools/ubench/bin/neighbour
mkdir -p tools/ubench/bin
[ -x $N ] || /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/ubench/neighbour.hip -o $N
for spec in "$@"; do
  $N $KIND 60000 2 32 > /dev/null & P1=$!
  $N $KIND 60000 2 32 > /dev/null & P2=$!
  echo "=== 2 x $KIND beside $spec"
  timeout 150 python tools/stress_determinism.py --mix "$spec" --reps $REPS --n 20000 2>&1 | grep -v "^      rep" | cut -c1-330
  kill $P1 $P2 2>/dev/null; wait $P1 $P2 2>/dev/null
done
