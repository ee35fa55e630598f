set -e
cd $GRAFT_REPO_ROOT
for v in "-DAL_NOSB" "-DAL_NOSB -DAL_ABL=1" "-DAL_NOSB -DAL_ABL=2" "-DAL_NOSB -DAL_ABL=3"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $v -c video-k-net_amd/csrc/vkn_assign_lr.hip -o video-k-net_amd/lib/obj/vkn_assign_lr.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC video-k-net_amd/lib/obj/*.o -o video-k-net_amd/lib/libvkn.so
  echo "== $v"
  python tools/assign_lr_time.py 2>&1 | grep "low-res"
done
