// vmcnt_order_probe: does gfx950 retire vector-memory LOADS and STORES of one wave in issue order on the single vmcnt counter?
// hipcc's waitcnt pass assumes it (gfx9 has no separate store counter): behind `load A; store B1..Bn; use A` it emits s_waitcnt vmcnt(n).
// The probe issues, per wave, one load that misses every cache (a fresh 128-B line of a large buffer), then n stores to a line the wave
// has already written (L2-resident), waits with vmcnt(n) and copies the load's destination register at once.  A destination that still
// holds the sentinel means a store was counted as complete before the older load had returned: counted waits across loads and stores
// are then unsafe on this chip.  Round 5 (tools/probe/README or DESIGN 3.3): k_sib_children2 with its loads moved in front of its stores
// and counted waits was NOT reproducible run to run; this is the isolated question.
// usage: vmcnt_order_probe [stores per pass = 12] [passes = 64]   -> prints the number of stale reads
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int NST>
__global__ __launch_bounds__(64) void k_probe(const uint32_t* __restrict__ cold, uint32_t* __restrict__ hot, uint32_t* __restrict__ stale, int passes, size_t stride_words) {
    const int lane = threadIdx.x;
    const size_t wave = blockIdx.x;
    uint32_t* myhot = hot + wave * 64 * 16 + lane; // a few lines this wave keeps rewriting
    unsigned bad = 0;
    for (int p = 0; p < passes; ++p) {
        const uint32_t* src = cold + ((wave * passes + p) * stride_words) + lane; // a line nobody has touched
        uint32_t r = 0xDEADBEEFu, got;
        if (NST == 12)
            asm volatile("global_load_dword %0, %2, off\n\t"
                         "global_store_dword %3, %4, off\n\tglobal_store_dword %3, %4, off offset:256\n\tglobal_store_dword %3, %4, off offset:512\n\t"
                         "global_store_dword %3, %4, off offset:768\n\tglobal_store_dword %3, %4, off offset:1024\n\tglobal_store_dword %3, %4, off offset:1280\n\t"
                         "global_store_dword %3, %4, off offset:1536\n\tglobal_store_dword %3, %4, off offset:1792\n\tglobal_store_dword %3, %4, off offset:2048\n\t"
                         "global_store_dword %3, %4, off offset:2304\n\tglobal_store_dword %3, %4, off offset:2560\n\tglobal_store_dword %3, %4, off offset:2816\n\t"
                         "s_waitcnt vmcnt(12)\n\t"
                         "v_mov_b32 %1, %0\n\t"
                         "s_waitcnt vmcnt(0)"
                         : "+v"(r), "=v"(got) : "v"(src), "v"(myhot), "v"((uint32_t)p) : "memory");
        else
            asm volatile("global_load_dword %0, %2, off\n\t"
                         "global_store_dword %3, %4, off nt\n\tglobal_store_dword %3, %4, off offset:256 nt\n\tglobal_store_dword %3, %4, off offset:512 nt\n\t"
                         "global_store_dword %3, %4, off offset:768 nt\n\t"
                         "s_waitcnt vmcnt(4)\n\t"
                         "v_mov_b32 %1, %0\n\t"
                         "s_waitcnt vmcnt(0)"
                         : "+v"(r), "=v"(got) : "v"(src), "v"(myhot), "v"((uint32_t)p) : "memory");
        bad += got == 0xDEADBEEFu ? 1u : 0u;
    }
    if (bad) atomicAdd(stale, bad);
}

int main(int argc, char** argv) {
    const int nst = argc > 1 ? atoi(argv[1]) : 12, passes = argc > 2 ? atoi(argv[2]) : 64;
    const int waves = 256 * 16;
    const size_t stride_words = 1024; // 4 KiB between the lines a wave reads: every load is an HBM miss
    const size_t cold_words = (size_t)waves * passes * stride_words + 64;
    uint32_t *cold, *hot, *stale;
    if (hipMalloc(&cold, cold_words * 4) != hipSuccess || hipMalloc(&hot, (size_t)waves * 64 * 16 * 4 + 4096) != hipSuccess || hipMalloc(&stale, 4) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(cold, 0x11, cold_words * 4); // (never equals the sentinel)
    hipMemset(hot, 0, (size_t)waves * 64 * 16 * 4 + 4096);
    unsigned total = 0;
    for (int rep = 0; rep < 8; ++rep) {
        hipMemset(stale, 0, 4);
        if (nst == 12) k_probe<12><<<waves, 64>>>(cold, hot, stale, passes, stride_words);
        else k_probe<4><<<waves, 64>>>(cold, hot, stale, passes, stride_words);
        unsigned h = 0;
        if (hipMemcpy(&h, stale, 4, hipMemcpyDeviceToHost) != hipSuccess) { printf("kernel failed: %s\n", hipGetErrorString(hipGetLastError())); return 1; }
        total += h;
        printf("rep %d: %u stale reads of %llu (load; %d stores%s; s_waitcnt vmcnt(%d); read)\n", rep, h, (unsigned long long)waves * passes * 64, nst == 12 ? 12 : 4, nst == 12 ? "" : " nt", nst == 12 ? 12 : 4);
    }
    printf("%s\n", total ? "OUT OF ORDER: a store was counted as complete before an older load had returned" : "in order: no stale read");
    return 0;
}
