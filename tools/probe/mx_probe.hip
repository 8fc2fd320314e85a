// Probe of the gfx950 block-scaled MFMA operand layout (no ISA doc in this image): feeds one-hot fp8 operands and
// prints which (row, col, k) each lane element maps to.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

// A: every lane/byte encodes a distinct value? fp8 cannot hold many distinct values, so instead run many launches:
// launch (la, ja): A has 1.0 only at lane la byte ja; B = all ones -> D[row][*] = 1 tells the row of that element.
// launch (lb, jb) with A = all ones, B one-hot -> the column.  k pairing: A one-hot (la,ja), B one-hot (lb,jb): D != 0 iff same k.
__global__ void probe(const uint8_t* a_bytes, const uint8_t* b_bytes, float* out, int scale_a, int scale_b) {
    const int lane = threadIdx.x;
    v8i a, b;
    for (int i = 0; i < 8; ++i) {
        a[i] = ((const int*)a_bytes)[lane * 8 + i];
        b[i] = ((const int*)b_bytes)[lane * 8 + i];
    }
    v16f c = {0};
    // cbsz = 0 (A fp8 e4m3), blgp = 0 (B fp8 e4m3), opsel 0
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, scale_a, 0, scale_b);
    for (int i = 0; i < 16; ++i) out[lane * 16 + i] = c[i];
}

int main() {
    uint8_t *da, *db; float* dout;
    hipMalloc(&da, 64 * 32); hipMalloc(&db, 64 * 32); hipMalloc(&dout, 64 * 16 * 4);
    std::vector<uint8_t> ha(2048), hb(2048);
    std::vector<float> ho(1024);
    const uint8_t ONE = 0x38; // e4m3 1.0
    auto run = [&](int sa, int sb) {
        hipMemcpy(da, ha.data(), 2048, hipMemcpyHostToDevice);
        hipMemcpy(db, hb.data(), 2048, hipMemcpyHostToDevice);
        probe<<<1, 64>>>(da, db, dout, sa, sb);
        hipMemcpy(ho.data(), dout, 4096, hipMemcpyDeviceToHost);
    };
    auto D = [&](int row, int col) { // documented C/D map: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
        for (int l = 0; l < 64; ++l) for (int r = 0; r < 16; ++r)
            if ((l & 31) == col && ((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) == row) return ho[l * 16 + r];
        return -1.0f;
    };
    // 1. scale semantics: all ones, K = 64 -> expect 64 * 2^(sa-127) * 2^(sb-127)
    for (auto& x : ha) x = ONE; for (auto& x : hb) x = ONE;
    int scs[4][2] = {{127, 127}, {128, 127}, {127, 126}, {0x7f7f7f7f, 0x7f7f7f7f}};
    for (auto& s : scs) { run(s[0], s[1]); printf("scale_a=%#x scale_b=%#x -> D[0][0]=%g D[5][7]=%g D[31][31]=%g\n", s[0], s[1], D(0, 0), D(5, 7), D(31, 31)); }
    // 2. rows of A elements: A one-hot at (lane, byte), B all ones
    printf("A element (lane,byte) -> row\n");
    for (int la : {0, 1, 31, 32, 33, 63}) for (int ja : {0, 1, 15, 16, 31}) {
        for (auto& x : ha) x = 0; ha[la * 32 + ja] = ONE; for (auto& x : hb) x = ONE;
        run(127, 127);
        int row = -1, cnt = 0; for (int r = 0; r < 32; ++r) if (D(r, 0) != 0) { row = r; ++cnt; }
        printf("  A(%d,%d) -> row %d (rows hit %d, val %g)\n", la, ja, row, cnt, row >= 0 ? D(row, 0) : 0.f);
    }
    printf("B element (lane,byte) -> col\n");
    for (int lb : {0, 1, 31, 32, 33, 63}) for (int jb : {0, 1, 15, 16, 31}) {
        for (auto& x : hb) x = 0; hb[lb * 32 + jb] = ONE; for (auto& x : ha) x = ONE;
        run(127, 127);
        int col = -1, cnt = 0; for (int c = 0; c < 32; ++c) if (D(0, c) != 0) { col = c; ++cnt; }
        printf("  B(%d,%d) -> col %d (cols hit %d)\n", lb, jb, col, cnt);
    }
    // 3. k pairing: A one-hot (lane la in row 0.., byte ja); find which B (lane with col 0, byte) matches
    printf("k pairing: A(lane,byte) matches B(lane,byte)\n");
    for (int la : {0, 32}) for (int ja : {0, 1, 3, 4, 15, 16, 17, 31}) {
        for (auto& x : ha) x = 0; ha[la * 32 + ja] = ONE;
        for (int lb : {0, 32}) for (int jb = 0; jb < 32; ++jb) {
            for (auto& x : hb) x = 0; hb[lb * 32 + jb] = ONE;
            run(127, 127);
            if (D(0, 0) != 0) printf("  A(%d,%d) <-> B(%d,%d)\n", la, ja, lb, jb);
        }
    }
    return 0;
}
