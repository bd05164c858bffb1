# The 1x1 GEMM on the ResNet-50 shapes at ring depth 1 / 2 / 3 (gpurun).  Round 3 also ran round 2's register-staged kernel
# through this script (since removed); its numbers are in profiles/r03_exp_gemm_sweep.txt.
cd $GRAFT_REPO_ROOT
for v in 1 2 3; do DH_G2_NSTAGE=$v python tools/gemm_bench.py gemm > gpurun_out/r3_gs_$v.txt 2>&1; done
paste gpurun_out/r3_gs_1.txt gpurun_out/r3_gs_2.txt gpurun_out/r3_gs_3.txt | awk '{print $1,$2,$3,"ns1",$4,"ns2",$12,"ns3",$20}'
