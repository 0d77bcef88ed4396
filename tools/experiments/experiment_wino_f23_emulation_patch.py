# experiment patch (round 3): Winograd F(2,3) along W emulated inside conv_sf16_kernel (timing only, results wrong): build with -DSF_EMU_WINO;
# bash tools/ablate.sh conv_sf16.hip "--batch 128 --reps 5 --only sf16" base DSF_EMU_WINO
# Applies to csrc/conv_sf16.hip as of commit 09bea50 (git show 09bea50:sound_event_detection_dcase2017_task4_amd/csrc/conv_sf16.hip);
# the string anchors below fail loudly on any other revision.  Result: docs/HISTORY.md (round 3), profiles/r03/experiment_*.txt.
p='/root/repo/sound_event_detection_dcase2017_task4_amd/csrc/conv_sf16.hip'
s=open(p).read()
def rep(a,b,cnt=1):
    global s
    assert s.count(a)==cnt, (a, s.count(a))
    s=s.replace(a,b)
rep("__global__ __launch_bounds__(256, MW == 4 ? 3 : 2) void conv_sf16_kernel(Sf16P p) {",
"""#ifdef SF_EMU_WINO
__global__ __launch_bounds__(256, 2) void conv_sf16_kernel(Sf16P p) {
#else
__global__ __launch_bounds__(256, MW == 4 ? 3 : 2) void conv_sf16_kernel(Sf16P p) {
#endif""")
rep("""    constexpr int AROWS = MW == 2 ? 264 : 408;         // >= (TR+2) * WP over the supported W (W = 8: 34 * 12)
    constexpr int APLANE = AROWS * 32;
    constexpr int BPLANE = 3 * BN * 32, BSTAGE = 2 * BPLANE;
    constexpr int NI = MW == 2 ? 4 : 6;                // staging items per thread
    constexpr int NDMA = 6 * RB / 4;                   // LDS-DMA instructions per wave and stage
""","""#ifdef SF_EMU_WINO
    constexpr int AROWS = 768, NPOS = 4;
#else
    constexpr int AROWS = MW == 2 ? 264 : 408;         // >= (TR+2) * WP over the supported W (W = 8: 34 * 12)
    constexpr int NPOS = 3;
#endif
    constexpr int APLANE = AROWS * 32;
    constexpr int BPLANE = NPOS * BN * 32, BSTAGE = 2 * BPLANE;
    constexpr int NI = MW == 2 ? 4 : 6;                // staging items per thread
    constexpr int NDMA = 2 * NPOS * RB / 4;            // LDS-DMA instructions per wave and stage
""")
rep("""    float4 areg##i = make_float4(0.f, 0.f, 0.f, 0.f);                                                           \\
    if (i < NI) {""","""    float4 areg##i = make_float4(0.f, 0.f, 0.f, 0.f), breg##i = areg##i;                                        \\
    if (i < NI) {""")
rep("""#define SF_ALOAD(i) if (i < NI) areg##i = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(xrs, aoff##i, k_off, 0));""",
"""#ifdef SF_EMU_WINO
#define SF_ALOAD(i) if (i < NI) { areg##i = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(xrs, aoff##i, k_off, 0)); \\
                                  breg##i = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(xrs, aoff##i ^ 16, k_off, 0)); }
#else
#define SF_ALOAD(i) if (i < NI) areg##i = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(xrs, aoff##i, k_off, 0));
#endif""")
rep("""        *reinterpret_cast<uint2*>(As + lso##i) = make_uint2(h01, h23);                                          \\
        *reinterpret_cast<uint2*>(As + APLANE + lso##i) = make_uint2(l01, l23);                                 \\
    }
#define sf_astore()""","""        *reinterpret_cast<uint2*>(As + lso##i) = make_uint2(h01, h23);                                          \\
        *reinterpret_cast<uint2*>(As + APLANE + lso##i) = make_uint2(l01, l23);                                 \\
        SF_EMU_SECOND(i)                                                                                        \\
    }
#define sf_astore()""")
rep("""#define SF_ASTORE(i)                                                                                            \\
    if (i < NI && (INT ? sok##i : val##i)) {""","""#ifdef SF_EMU_WINO
#define SF_EMU_SECOND(i)                                                                                        \\
        {                                                                                                       \\
            float4 u = breg##i;                                                                                 \\
            if (INT) {                                                                                          \\
                u.x = bn_relu(u.x, sc4.x, sh4.x); u.y = bn_relu(u.y, sc4.y, sh4.y);                             \\
                u.z = bn_relu(u.z, sc4.z, sh4.z); u.w = bn_relu(u.w, sc4.w, sh4.w);                             \\
            } else {                                                                                            \\
                u.x *= sa; u.y *= sa; u.z *= sa; u.w *= sa;                                                     \\
            }                                                                                                   \\
            const float4 d = make_float4(v.x - u.x, v.y - u.y, v.z - u.z, v.w - u.w);                           \\
            const float4 e2 = make_float4(0.5f * (v.x + u.x), 0.5f * (v.y + u.y), 0.5f * (v.z + u.z), 0.5f * (v.w + u.w)); \\
            overflow |= !((fabsf(d.x) + fabsf(d.y)) + (fabsf(d.z) + fabsf(d.w)) < 3.0e5f);                      \\
            sf_split2(d.x, d.y, h01, l01);                                                                      \\
            sf_split2(d.z, d.w, h23, l23);                                                                      \\
            *reinterpret_cast<uint2*>(As + 360 * 32 + lso##i) = make_uint2(h01, h23);                           \\
            *reinterpret_cast<uint2*>(As + APLANE + 360 * 32 + lso##i) = make_uint2(l01, l23);                  \\
            sf_split2(e2.x, e2.y, h01, l01);                                                                    \\
            sf_split2(e2.z, e2.w, h23, l23);                                                                    \\
            *reinterpret_cast<uint2*>(As + lso##i) = make_uint2(h01, h23);                                      \\
            *reinterpret_cast<uint2*>(As + APLANE + lso##i) = make_uint2(l01, l23);                             \\
        }
#else
#define SF_EMU_SECOND(i)
#endif
#define SF_ASTORE(i)                                                                                            \\
    if (i < NI && (INT ? sok##i : val##i)) {""")
rep("""        const int pl = qi / (3 * RB), dxx = (qi / RB) % 3, rb = qi % RB;                                        \\
        const unsigned char* src = reinterpret_cast<const unsigned char*>(p.wp) + (long)(STEP) * b_step_stride + \\
                                   pl * b_plane_stride + dxx * b_dx_stride + rb * 32 * 32;                      \\""",
"""        const int pl = qi / (NPOS * RB), dxx = (qi / RB) % NPOS, rb = qi % RB;                                  \\
        const unsigned char* src = reinterpret_cast<const unsigned char*>(p.wp) + (long)(STEP) * b_step_stride + \\
                                   pl * b_plane_stride + (dxx % 3) * b_dx_stride + rb * 32 * 32;                \\""")
rep("""    int aoffs[2][9], boffs[2][3];""","""    int aoffs[2][9], boffs[2][NPOS];""")
rep("""        for (int dx = 0; dx < 3; ++dx) boffs[nk][dx] = sf_sw(dx * BN + row, kh);""","""        for (int dx = 0; dx < NPOS; ++dx) boffs[nk][dx] = sf_sw(dx * BN + row, kh);""")
rep("""    floatx16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)""","""#ifdef SF_EMU_WINO
    floatx16 accw[4][2];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) accw[a][c][r] = 0.f;
#endif
    floatx16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)""")
rep("""#else
#ifdef SF_ABL_FRAGONCE     // timing experiment""","""#elif defined(SF_EMU_WINO)
#pragma unroll
            for (int pp = 0; pp < 2; ++pp) {       // two Winograd positions at a time: four independent accumulators
                half8 ah[2], al[2], bh[2][2], bl[2][2];
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int pos = pp * 2 + q;
                    const int ao = aoffs[pos & 1][dy * 3 + (pos >> 1)] + (pos == 3 ? 360 * 32 : 0);
                    ah[q] = *reinterpret_cast<const half8*>(As + ao);
                    al[q] = *reinterpret_cast<const half8*>(As + APLANE + ao);
#pragma unroll
                    for (int nk = 0; nk < 2; ++nk) {
                        bh[q][nk] = *reinterpret_cast<const half8*>(Bst + boffs[nk][pos]);
                        bl[q][nk] = *reinterpret_cast<const half8*>(Bst + BPLANE + boffs[nk][pos]);
                    }
                }
#pragma unroll
                for (int q = 0; q < 2; ++q)
#pragma unroll
                    for (int nk = 0; nk < 2; ++nk)
                        accw[pp * 2 + q][nk] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[q], bh[q][nk], accw[pp * 2 + q][nk], 0, 0, 0);
#pragma unroll
                for (int q = 0; q < 2; ++q)
#pragma unroll
                    for (int nk = 0; nk < 2; ++nk)
                        accw[pp * 2 + q][nk] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[q], bl[q][nk], accw[pp * 2 + q][nk], 0, 0, 0);
#pragma unroll
                for (int q = 0; q < 2; ++q)
#pragma unroll
                    for (int nk = 0; nk < 2; ++nk)
                        accw[pp * 2 + q][nk] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[q], bh[q][nk], accw[pp * 2 + q][nk], 0, 0, 0);
            }
#else
#ifdef SF_ABL_FRAGONCE     // timing experiment""")
rep("""    // ---- epilogue: unscale, (mask,) statistics, store (rows past the image fall outside the descriptor and are dropped)
""","""#ifdef SF_EMU_WINO
#pragma unroll
    for (int nk = 0; nk < 2; ++nk)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            acc[0][nk][r] = accw[0][nk][r] + accw[1][nk][r] + accw[2][nk][r];
            acc[1][nk][r] = accw[1][nk][r] - accw[2][nk][r] - accw[3][nk][r];
        }
#endif
    // ---- epilogue: unscale, (mask,) statistics, store (rows past the image fall outside the descriptor and are dropped)
""")
open(p,'w').write(s)
print("ok")
