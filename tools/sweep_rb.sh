for rb in 64 128 256; do
ZG_MSM_RB=$rb python bench.py --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null > gpurun_out/rb.json; python -c "
import json; d=json.load(open('gpurun_out/rb.json')); k=d['single_proof_kernels_ms']; print($rb, round(d['ms_per_proof'],3), round(d['create_proof_wall_s']*1e3,3), d['single_proof_phase_ms']['total'], {n:k[n] for n in ('msm_bucket_scan','msm_bucket_sum','msm_finish')})"
done
