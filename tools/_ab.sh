for i in 1 2 3; do
for lib in libccsm_old.so libccsm.so; do
CCSM_LIB_PATH=$GRAFT_REPO_ROOT/ccsmeth_amd/lib/$lib python bench.py 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_ms']; print('$lib', round(d['value']), round(k['gru0'],3), round(k['gru1'],3), round(k['gru2'],3), round(k['attn_fc'],3))"
done; done
