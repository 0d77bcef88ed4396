# experiment patch (round 3): bn_bwd_apply folded into the staging of the dgrad / weight-gradient kernels (second tensor + per-channel affine;
# timing only, results wrong): rebuild and run tools/conv_bench.py --only sf16 / --only sf16w
# Applies to csrc/conv_sf16.hip as of commit 59f285e (git show 59f285e:sound_event_detection_dcase2017_task4_amd/csrc/conv_sf16.hip);
# the string anchors below fail loudly on any other revision.  Result: docs/HISTORY.md (round 3), profiles/r03/experiment_*.txt.
p='/root/repo/sound_event_detection_dcase2017_task4_amd/csrc/conv_sf16.hip'
s=open(p).read()
def rep(a,b,cnt=1):
    global s
    assert s.count(a)==cnt,(a[:70],s.count(a))
    s=s.replace(a,b)
# ---- conv: non-INT staging gets a second tensor + per-channel affine (timing emulation: results wrong)
rep("""    float4 areg##i = make_float4(0.f, 0.f, 0.f, 0.f);                                                           \\
    if (i < NI) {""","""    float4 areg##i = make_float4(0.f, 0.f, 0.f, 0.f), breg##i = areg##i;                                        \\
    if (i < NI) {""")
rep("""    const __amdgpu_buffer_rsrc_t srs = __builtin_amdgcn_make_buffer_rsrc(""","""    const __amdgpu_buffer_rsrc_t xrs2 = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.x) + (long)((b + p.B / 2) % p.B) * p.H * W * p.K, 0, (int)((unsigned)p.H * W * p.K * 4u), 0x00020000);
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.wscale), 0, 256, 0x00020000);
    float4 fa4 = make_float4(1.f, 1.f, 1.f, 1.f), fb4 = fa4, fc4 = fa4;
    const __amdgpu_buffer_rsrc_t srs = __builtin_amdgcn_make_buffer_rsrc(""")
rep("""#define SF_ALOAD(i) if (i < NI) areg##i = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(xrs, aoff##i, k_off, 0));""",
"""#define SF_ALOAD(i) if (i < NI) { areg##i = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(xrs, aoff##i, k_off, 0)); \\
                                  if (!INT) breg##i = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(xrs2, aoff##i, k_off, 0)); }""")
rep("""        SF_ALOAD(0) SF_ALOAD(1) SF_ALOAD(2) SF_ALOAD(3) SF_ALOAD(4) SF_ALOAD(5)                                 \\
    }""","""        if (!INT) {                                                                                             \\
            fa4 = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(wrs, q4 * 16, 0, 0));        \\
            fb4 = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(wrs, q4 * 16, 64, 0));       \\
            fc4 = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(wrs, q4 * 16, 128, 0));      \\
            fa4.x *= sa; fa4.y *= sa; fa4.z *= sa; fa4.w *= sa; fb4.x *= 1e-9f; fb4.y *= 1e-9f; fb4.z *= 1e-9f; fb4.w *= 1e-9f; \\
            fc4.x *= 1e-9f; fc4.y *= 1e-9f; fc4.z *= 1e-9f; fc4.w *= 1e-9f;                                     \\
        }                                                                                                       \\
        SF_ALOAD(0) SF_ALOAD(1) SF_ALOAD(2) SF_ALOAD(3) SF_ALOAD(4) SF_ALOAD(5)                                 \\
    }""")
rep("""        } else {                                                                                                \\
            v.x *= sa; v.y *= sa; v.z *= sa; v.w *= sa;        /* out-of-image loads returned 0 */              \\
        }""","""        } else {                                                                                                \\
            const float4 u = breg##i;                                                                           \\
            v.x = fmaf(fa4.x, v.x, fmaf(fb4.x, u.x, fc4.x)); v.y = fmaf(fa4.y, v.y, fmaf(fb4.y, u.y, fc4.y));   \\
            v.z = fmaf(fa4.z, v.z, fmaf(fb4.z, u.z, fc4.z)); v.w = fmaf(fa4.w, v.w, fmaf(fb4.w, u.w, fc4.w));   \\
        }""")
# ---- wgrad: gy staging gets a second tensor + affine
rep("""    float4 xreg[2], greg[4];""","""    float4 xreg[2], greg[4], hreg[4];
    const float4 ga4 = *reinterpret_cast<const float4*>(p.g_amax + (tid & 3) * 4);""")
rep("""    grs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.gy) + (long)(BB) * p.H * W * p.N, 0, (int)g_img_bytes, 0x00020000);""",
"""    grs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.gy) + (long)(BB) * p.H * W * p.N, 0, (int)g_img_bytes, 0x00020000); \\
    grs2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.gy) + (long)(((BB) + p.B / 2) % p.B) * p.H * W * p.N, 0, (int)g_img_bytes, 0x00020000);""")
rep("""    __amdgpu_buffer_rsrc_t xrs, grs;""","""    __amdgpu_buffer_rsrc_t xrs, grs, grs2;""")
rep("""        greg[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(grs, goff[i], (H0) * W * p.N * 4, 0));""",
"""        { greg[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(grs, goff[i], (H0) * W * p.N * 4, 0)); \\
          hreg[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(grs2, goff[i], (H0) * W * p.N * 4, 0)); }""")
rep("""        float4 v = greg[i];                                                                                     \\
        v.x *= sg; v.y *= sg; v.z *= sg; v.w *= sg;                                                             \\""",
"""        float4 v = greg[i];                                                                                     \\
        { const float4 u = hreg[i];                                                                             \\
          v.x = fmaf(sg, v.x, fmaf(ga4.x * 1e-12f, u.x, ga4.y * 1e-12f)); v.y = fmaf(sg, v.y, fmaf(ga4.y * 1e-12f, u.y, ga4.z * 1e-12f)); \\
          v.z = fmaf(sg, v.z, fmaf(ga4.z * 1e-12f, u.z, ga4.w * 1e-12f)); v.w = fmaf(sg, v.w, fmaf(ga4.w * 1e-12f, u.w, ga4.x * 1e-12f)); } \\""")
open(p,'w').write(s)
print("ok")
