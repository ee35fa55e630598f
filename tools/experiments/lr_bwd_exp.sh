set -e
cd $GRAFT_REPO_ROOT
for lb in 2 3; do
  sed "s/__launch_bounds__(64 \* LR_NW, 2) void k_ml_bwd_lr/__launch_bounds__(64 * LR_NW, $lb) void k_ml_bwd_lr/" video-k-net_amd/csrc/vkn_loss.hip > video-k-net_amd/csrc/_exp_loss.hip
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c video-k-net_amd/csrc/_exp_loss.hip -o video-k-net_amd/lib/obj/vkn_loss.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC video-k-net_amd/lib/obj/*.o -o video-k-net_amd/lib/libvkn.so
  echo "== bound $lb"
  python tools/lr_bwd_time.py 2>&1 | grep "low-res\|max"
done
rm -f video-k-net_amd/csrc/_exp_loss.hip
