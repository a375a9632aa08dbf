O=gpurun_out/${1:-r03hl}
mkdir -p $O
for b in 16 64 16 64; do
  echo "PT_SHIM_BATCH=$b"; PT_SHIM_BATCH=$b project3-pathtracer_amd/lib/pt_headless scene=scenes/sampleScene_spec.txt res=1920x1080 iterations=2048 out=$O 2>&1 | tail -2
done
