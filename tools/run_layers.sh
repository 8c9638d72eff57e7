L=$GRAFT_REPO_ROOT/scenedreamer_amd/lib
cp $L/libsdnative.so /tmp/orig.so; cp $L/variants/ablation.so $L/libsdnative.so
for a in "512 two" "512 one"; do echo "=== SDN_MLP_DBG / field: $a"; timeout 300 python tools/dbg_layers.py $a 2>&1 | grep -vE "Warning|warn|amdgpu.ids"; done
cp /tmp/orig.so $L/libsdnative.so
