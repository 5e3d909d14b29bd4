for n in tree stag2 stag5; do
  lib=$PWD/figdraw_amd/libfigdraw_hip.so; [ $n != tree ] && lib=$PWD/build/libfigdraw_hip_$n.so
  echo "== $n"; FIGDRAW_HIP_LIB=$lib python tools/blur_t_sweep.py 2>&1 | sed -n '1p;4p'
done
