# the C++ client of the C ABI (no Python): modes 0-4, mode 4 = per-view calls chained across two probed streams
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "c_abi_client" > gpurun_out/r05_cclient.log 2>&1; rc=$?
tail -3 gpurun_out/r05_cclient.log
hipcc --offload-arch=gfx950 -O2 -std=c++17 -Iinclude tests/c_client/abi_client.cpp -Ldepthdensifier_amd -lddcore -Wl,-rpath,$PWD/depthdensifier_amd -o /tmp/abi_client && timeout -k 10 120 /tmp/abi_client | tee gpurun_out/r05_cclient_out.txt | tail -12
exit $rc
