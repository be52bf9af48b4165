mkdir -p gpurun_out
export TCNN_HIP_LIBRARY=$PWD/tiny-cuda-nn_amd/lib/variants/w2.so
for pair in "None ReLU" "ReLU None" "None None"; do
set -- $pair
TCNN_MLP_TRAIN_WAVE=1 python scripts/dbg_wave.py $1 $2 wave_$1_$2 2>&1 | tail -1
TCNN_MLP_TRAIN_WAVE=0 python scripts/dbg_wave.py $1 $2 tiled_$1_$2 2>&1 | tail -1
python - <<PY
import numpy as np
a=np.load("gpurun_out/dbg_wave_$1_$2.npz"); b=np.load("gpurun_out/dbg_tiled_$1_$2.npz")
nm=7168
d=np.abs(a["g"][:nm]-b["g"][:nm]); print("$1 $2 mlp grad max diff", d.max(), "rel", d.max()/np.abs(b["g"][:nm]).max(), "grid equal", np.array_equal(a["g"][nm:], b["g"][nm:]))
for name,lo,hi in (("W_in",0,2048),("W_hid",2048,6144),("W_out",6144,7168)):
    print("  ", name, np.abs(a["g"][lo:hi]-b["g"][lo:hi]).max(), np.abs(b["g"][lo:hi]).max())
PY
done
