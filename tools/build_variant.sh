# build_variant.sh NAME "-DRFX_LT=768 -DRFX_LEAF_WAVES_PER_EU=6": libreflexiv_hip.so with extra flags on rfx_kmer.hip
# -> reflexiv_amd/lib_NAME.so.bak (git-ignored; travels with gpurun for tools/ab_many.sh)
set -e
cd "$(dirname "$0")/../reflexiv_amd/csrc"
name=$1; shift
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-unused-result $* \
    -Rpass-analysis=kernel-resource-usage -c rfx_kmer.hip -o /tmp/rfx_kmer_$name.o 2> /tmp/rfx_kmer_$name.res
objs=$(ls *.o | grep -v rfx_kmer.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib_$name.so.bak $objs /tmp/rfx_kmer_$name.o
grep -A12 "${KERN:-k_leaf_countILi1ELi31}" /tmp/rfx_kmer_$name.res | grep -E "VGPRs:|AGPRs|Spill|ScratchSize|Occupancy|LDS Size" | head -8
