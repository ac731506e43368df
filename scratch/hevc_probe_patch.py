"""Developer tool (round 5): writes an instrumented copy of jmcodec_amd/csrc/hevc_kernels.hip (wall-clock probes around the phases of k_hevc_intra's CTB loop
and block loop; totals printed at process exit) for a variant build -- `python scratch/hevc_probe_patch.py; make -C jmcodec_amd/csrc OUT=../lib_dbg_probe
OBJ=../lib_dbg_probe/obj; git checkout jmcodec_amd/csrc/hevc_kernels.hip`.  The product never contains the probes."""
import sys
p = 'jmcodec_amd/csrc/hevc_kernels.hip'
s = open(p).read()
def rep(old, new):
    global s
    assert s.count(old) == 1, (s.count(old), old[:80])
    s = s.replace(old, new)
rep('__constant__ uint8_t c_beta[52], c_tc[54], c_qpc[58];', '''__constant__ uint8_t c_beta[52], c_tc[54], c_qpc[58];
__device__ unsigned long long g_probe[32];
#define PROBE(i, t0) do { if (threadIdx.x == 0) { unsigned long long t1_ = wall_clock64(); atomicAdd(&g_probe[i], t1_ - (t0)); (t0) = t1_; } } while (0)''')
rep('''    while (cx < c1) {
    const HevcCtb ctb = s_ctb[cx - c0];''', '''    unsigned long long tp = wall_clock64();
    while (cx < c1) {
    PROBE(0, tp);
    const HevcCtb ctb = s_ctb[cx - c0];''')
rep('''    __syncthreads();                                              // (also: everybody is done with the previous tile)
    const int n_tbs = (int)ctb.intra_count;''', '''    __syncthreads();                                              // (also: everybody is done with the previous tile)
    PROBE(1, tp);
    const int n_tbs = (int)ctb.intra_count;''')
rep('''    if (next_cx < c1) prefetch(next_cx, pre);
    __syncthreads();''', '''    if (next_cx < c1) prefetch(next_cx, pre);
    __syncthreads();
    PROBE(2, tp);''')
rep('''    }   // wave < 3
    __syncthreads();''', '''    }   // wave < 3
    if (threadIdx.x == 0) atomicAdd(&g_probe[8], 1ull);
    PROBE(3, tp);
    __syncthreads();
    PROBE(4, tp);''')
rep('''    prev_cx = cx;
    cx = next_cx;''', '''    PROBE(5, tp);
    prev_cx = cx;
    cx = next_cx;''')
rep('''        const int16_t *e = edge[0];
        if (!pcm) {''', '''        const int16_t *e = edge[0];
        unsigned long long tq = wall_clock64();
        if (threadIdx.x == 0) { atomicAdd(&g_probe[16], 1ull); atomicAdd(&g_probe[20 + log2 - 2], 1ull); }
        if (!pcm) {''')
rep('''            __builtin_amdgcn_wave_barrier();
            // ---- filtering (8.4.4.2.3) ----''', '''            __builtin_amdgcn_wave_barrier();
            PROBE(9, tq);
            // ---- filtering (8.4.4.2.3) ----''')
rep('''        const int16_t *L = e + 2 * n - 1, *T = e + 2 * n + 1;''', '''        PROBE(10, tq);
        const int16_t *L = e + 2 * n - 1, *T = e + 2 * n + 1;''')
rep('''        // ref[i] == rbase[rs * i]''', '''        PROBE(11, tq);
        // ref[i] == rbase[rs * i]''')
rep('''        __builtin_amdgcn_wave_barrier();
    }
    }   // wave < 3''', '''        __builtin_amdgcn_wave_barrier();
        PROBE(12 + (log2 - 2), tq);
    }
    }   // wave < 3''')
rep('''        hipLaunchKernelGGL(k_hevc_intra, dim3(m.max_ctb_h * kHevcIntraSegs, n), dim3(kIntraThreads), 0, st, d_pics, progress, kHevcProgressStride);''',
'''        hipLaunchKernelGGL(k_hevc_intra, dim3(m.max_ctb_h * kHevcIntraSegs, n), dim3(kIntraThreads), 0, st, d_pics, progress, kHevcProgressStride);
        { static bool reg = false;
          if (!reg) { reg = true; atexit([] { unsigned long long h2[32]; hipMemcpyFromSymbol(h2, HIP_SYMBOL(g_probe), sizeof h2);
              fprintf(stderr, "HEVC_PROBE ctbs %llu tbs(wave0) %llu by size 4:%llu 8:%llu 16:%llu 32:%llu\\n", h2[8], h2[16], h2[20], h2[21], h2[22], h2[23]);
              const char *nm[16] = {"loop_top", "ctbrec+wait_above", "records+row_above+fill+prefetch_issue", "block_loop(wave0)", "wait_other_planes", "bottom_store+publish+rest", "", "", "", "tb_gather", "tb_filter", "tb_dc_refa", "tb_predict_4", "tb_predict_8", "tb_predict_16", "tb_predict_32"};
              for (int i = 0; i < 16; i++) if (nm[i][0]) { const double per = i < 8 ? (double)(h2[8] ? h2[8] : 1) : (i >= 12 ? (double)(h2[20 + i - 12] ? h2[20 + i - 12] : 1) : (double)(h2[16] ? h2[16] : 1));
                  fprintf(stderr, "HEVC_PROBE %-40s total %10.1f us  per %s %7.3f us\\n", nm[i], h2[i] / 100.0, i < 8 ? "ctb" : "tb", h2[i] / 100.0 / per); }
          }); } }''')
open(p, 'w').write(s)
