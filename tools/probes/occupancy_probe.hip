// How many wavefronts of a kernel with a given VGPR count does one SIMD of gfx950 really hold?  (The compiler reports
// floor(512 / vgprs); the dispatcher of this chip was seen to hold ONE 5-wavefront workgroup of a 168-VGPR kernel per CU.)
// Each variant: 4096 workgroups of WAVES wavefronts that spin ~15 us; concurrency per CU from timestamps + HW_ID.
// build: hipcc -O2 --offload-arch=gfx950 occupancy_probe.hip -o occupancy_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <map>
#include <algorithm>
template <int VG>
__global__ void probe(long long* out, int lds_bytes)
{
    extern __shared__ char smem[];
    if (VG == 128) asm volatile("" ::: "v127"); if (VG == 136) asm volatile("" ::: "v135"); if (VG == 144) asm volatile("" ::: "v143");
    if (VG == 152) asm volatile("" ::: "v151"); if (VG == 160) asm volatile("" ::: "v159"); if (VG == 168) asm volatile("" ::: "v167");
    if (VG == 96) asm volatile("" ::: "v95"); if (VG == 64) asm volatile("" ::: "v63");
    const long long t0 = wall_clock64();
    if (lds_bytes > 0 && threadIdx.x == 0) smem[lds_bytes - 1] = 1;
    while (wall_clock64() - t0 < 1500) __builtin_amdgcn_s_sleep(8);
    if (threadIdx.x == 0) {
        out[blockIdx.x * 4 + 0] = t0; out[blockIdx.x * 4 + 1] = wall_clock64();
        out[blockIdx.x * 4 + 2] = __builtin_amdgcn_s_getreg(63492); out[blockIdx.x * 4 + 3] = __builtin_amdgcn_s_getreg(63508) & 15;
    }
}
template <int VG>
void run(int waves, int lds)
{
    const int nb = 4096;
    long long* d; hipMalloc(&d, sizeof(long long) * 4 * nb);
    hipFuncSetAttribute((const void*)probe<VG>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL(probe<VG>, dim3(nb), dim3(64 * waves), lds, 0, d, lds);
    hipDeviceSynchronize();
    std::vector<long long> h(4 * nb); hipMemcpy(h.data(), d, sizeof(long long) * 4 * nb, hipMemcpyDeviceToHost);
    // workgroups alive at the end time of the first workgroup to finish, per CU
    long long tend = h[1]; for (int b = 0; b < nb; ++b) tend = std::min(tend, h[b * 4 + 1]);
    std::map<long long, int> per;
    for (int b = 0; b < nb; ++b) if (h[b * 4] < tend) { const long long hw = h[b * 4 + 2]; per[(h[b * 4 + 3] << 16) | (((hw >> 13) & 7) << 8) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 15)]++; }
    int mx = 0; long tot = 0; for (auto& kv : per) { mx = std::max(mx, kv.second); tot += kv.second; }
    hipFuncAttributes fa; hipFuncGetAttributes(&fa, (const void*)probe<VG>);
    printf("vgprs asked %3d (compiler: %3d)  waves/wg %d  lds %6d B : %4ld workgroups resident on %3zu CUs, max %d per CU => %d waves per CU\n",
           VG, fa.numRegs, waves, lds, tot, per.size(), mx, mx * waves);
    hipFree(d);
}
int main()
{
    for (int waves : {1, 4, 5, 6}) {
        run<64>(waves, 0); run<96>(waves, 0); run<128>(waves, 0); run<136>(waves, 0); run<144>(waves, 0); run<152>(waves, 0); run<160>(waves, 0); run<168>(waves, 0);
    }
    run<64>(5, 51328); run<64>(6, 46328); run<64>(4, 6148);
    for (int lds : {23064, 22776, 22528, 22016, 21504, 20480}) run<64>(4, lds);    // the z passes of the 75 x 75 x 72 mesh: 6, 7 or 8 per CU?
    return 0;
}
