# what the box's driver does with queues of several processes (no GPU work): module parameters, KFD topology, kernel log if readable
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; o=gpurun_out/r06_box_info.txt
{
echo "== amdgpu module parameters"
for p in cwsr_enable sched_policy hws_max_conc_proc queue_preemption_timeout_ms mes mes_kiq mes_log_enable max_num_of_queues_per_device halt_if_hws_hang hws_gws_support noretry vm_fault_stop gpu_recovery reset_method debug_evictions no_queue_eviction_on_vm_fault num_kcq lockup_timeout mcbp sdma_phase_quantum svm_default_granularity; do
  f=/sys/module/amdgpu/parameters/$p; [ -r $f ] && echo "$p = $(cat $f)"; done
echo "== kfd topology node of the GPU"
for n in /sys/class/kfd/kfd/topology/nodes/*; do if grep -q "simd_count [1-9]" $n/properties 2>/dev/null; then echo $n; grep -E "simd_count|cu_count|max_waves|num_cp_queues|num_sdma|num_xcc|gfx_target|sdma_fw|fw_version|unique_id|debug_prop|capability|num_gws" $n/properties; fi; done
echo "== kernel / driver"; uname -r; cat /sys/module/amdgpu/version 2>/dev/null; ls /sys/kernel/debug/dri 2>&1 | head -3
echo "== dmesg (if readable)"; dmesg 2>&1 | tail -40
echo "== rocm-smi"; rocm-smi --showuse --showmemuse 2>&1 | head -20
echo "== processes on the GPU"; rocm-smi --showpids 2>&1 | head -20
} > $o 2>&1
cat $o | head -120
