# A/B of two builds on one box: tools/ab_old.so against the in-tree library.  gpurun -- 'bash tools/ab.sh <script> [rounds]'
S=$1; N=${2:-2}
cp mixlab_amd/libmixlab_gpu.so /tmp/ab_new.so
for i in $(seq $N); do
  cp tools/ab_old.so mixlab_amd/libmixlab_gpu.so; echo "== old"; bash $S
  cp /tmp/ab_new.so mixlab_amd/libmixlab_gpu.so; echo "== new"; bash $S
done
