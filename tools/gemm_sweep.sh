# A/B of the 1x1 GEMM variants on the ResNet-50 shapes (gpurun): old kernel, ring depth 1 / 2 / 3
cd $GRAFT_REPO_ROOT
for v in old 1 2 3; do
  if [ $v = old ]; then DH_GEMM_OLD=1 python tools/gemm_bench.py gemm > gpurun_out/r3_gs_$v.txt 2>&1
  else DH_G2_NSTAGE=$v python tools/gemm_bench.py gemm > gpurun_out/r3_gs_$v.txt 2>&1; fi
done
paste gpurun_out/r3_gs_old.txt gpurun_out/r3_gs_1.txt gpurun_out/r3_gs_2.txt gpurun_out/r3_gs_3.txt | awk '{print $1,$2,$3,"old",$4,"ns1",$12,"ns2",$20,"ns3",$28}'
