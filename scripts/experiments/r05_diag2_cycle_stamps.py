import re,sys
p='/root/repo/onephase.jl_amd/csrc/front_device.h'
s=open(p).read()
def rep(old,new,cnt=1):
    global s
    assert old in s, old[:60]
    s=s.replace(old,new,cnt)
rep('''  for (int ms = 0; ms < nms; ++ms) {
    const int p8 = ms * MW, pp = ms / PER, h = ms % PER;
    row_outputs();''','''#ifdef OKKT_D_CYC
  long long cy[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define CYC(i) do { if (marks && step == 0 && (tid == 64 || tid == 128)) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); cy[i] = clock64(); } } while (0)
#else
#define CYC(i) do {} while (0)
#endif
  for (int ms = 0; ms < nms; ++ms) {
    const int p8 = ms * MW, pp = ms / PER, h = ms % PER;
    row_outputs();
    if (ms == 7 && tid == 64) CYC(6);''')
# head: after operand reads / after MFMAs
i=s.index("      // head: update of step ms - 1 on the tile column of panel ms")
j=s.index("#pragma unroll\n      for (int q = 0; q < NT; ++q)\n        if (tj_s[q] == pp) {", i)
head=s[i:j]
head2=head.replace("#pragma unroll\n            for (int e = 0; e < NE; ++e) acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64","            if (ms == 7) CYC(3);\n#pragma unroll\n            for (int e = 0; e < NE; ++e) acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64")
head2+='''#ifdef OKKT_D_CYC
      if (ms == 7 && tid == 128) { asm volatile("" :: "v"(acc[0][0]), "v"(acc[1][0]), "v"(acc[2][0]), "v"(acc[3][0]), "v"(acc[4][0]), "v"(acc[5][0])); asm volatile("s_nop 7\\ns_nop 7" ::: "memory"); CYC(4); }
#endif
'''
s=s[:i]+head2+s[j:]
rep('''        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (marks && tid == 0) { const long long tn = wall_clock64(); t_head += tn - tk; tk = tn; }''','''        }
      if (ms == 7 && tid == 128) CYC(5);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (marks && tid == 0) { const long long tn = wall_clock64(); t_head += tn - tk; tk = tn; }
    if (ms == 6) CYC(0);
    if (ms == 7 && tid == 64) CYC(5);
    if (ms == 7 && tid == 128) CYC(6);''')
# row: after loads (both paths: before the first fast_rcp)
s=s.replace("#pragma unroll\n          for (int c = 0; c < MW; ++c) {\n            const double rdc = fast_rcp_f64(readlane_f64(a[c], c));","          if (ms == 6) CYC(1);\n#pragma unroll\n          for (int c = 0; c < MW; ++c) {\n            const double rdc = fast_rcp_f64(readlane_f64(a[c], c));")
s=s.replace("#pragma unroll\n          for (int c = 0; c < MW; ++c) {\n            rd[c] = fast_rcp_f64(A[c][c]);","          if (ms == 6) CYC(1);\n#pragma unroll\n          for (int c = 0; c < MW; ++c) {\n            rd[c] = fast_rcp_f64(A[c][c]);")
s=s.replace("#pragma unroll\n        for (int c = 0; c < MW; ++c) {\n          rd[c] = fast_rcp_f64(A[c][c]);","        if (ms == 6) CYC(1);\n#pragma unroll\n        for (int c = 0; c < MW; ++c) {\n          rd[c] = fast_rcp_f64(A[c][c]);")
# after solve / after panel writes
m=re.search(r"( *)if \(r >= p8\) \{\n#pragma unroll\n *for \(int c = 0; c < MW; \+\+c\) \{\n *Lp\[c \* kPLD \+ ppos\(r\)\] = -lr_keep\[c\];\n *Wp\[c \* kPLD \+ ppos\(r\)\] = w_keep\[c\];\n *\}\n *\}\n", s)
if m:
    blk=m.group(0)
    s=s.replace(blk,'#ifdef OKKT_D_CYC\n        if (ms == 6) { asm volatile("" :: "v"(a[MW - 1]), "v"(lr_keep[MW - 1])); CYC(2); }\n#endif\n'+blk+"        if (ms == 6) CYC(3);\n")
else:
    rep('''#pragma unroll
        for (int c = 0; c < MW; ++c) {
          Lp[c * kPLD + r] = -lr_keep[c];
          Wp[c * kPLD + r] = w_keep[c];
        }
''','''#ifdef OKKT_D_CYC
        if (ms == 6) { asm volatile("" :: "v"(a[MW - 1]), "v"(lr_keep[MW - 1])); CYC(2); }
#endif
#pragma unroll
        for (int c = 0; c < MW; ++c) {
          Lp[c * kPLD + r] = -lr_keep[c];
          Wp[c * kPLD + r] = w_keep[c];
        }
        if (ms == 6) CYC(3);
''')
# rest done + barrier2
i=s.index("      // rest of the update of step ms - 1")
j=s.index("    asm volatile(\"s_waitcnt lgkmcnt(0)\" ::: \"memory\");\n    if (marks && tid == 0) { const long long tn = wall_clock64(); t_row += tn - tk; tk = tn; }", i)
k=s.rindex("    }\n", i, j)
s=s[:k]+'''#ifdef OKKT_D_CYC
      if (ms == 6 && tid == 128) { asm volatile("" :: "v"(acc[0][0]), "v"(acc[1][0]), "v"(acc[2][0]), "v"(acc[3][0]), "v"(acc[4][0]), "v"(acc[5][0])); asm volatile("s_nop 7\\ns_nop 7" ::: "memory"); CYC(1); }
#endif
'''+s[k:]
rep('''    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (marks && tid == 0) { const long long tn = wall_clock64(); t_rest += tn - tk; tk = tn; }''','''    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (ms == 6 && tid == 64) CYC(4);
    if (ms == 6 && tid == 128) CYC(2);
    if (marks && tid == 0) { const long long tn = wall_clock64(); t_rest += tn - tk; tk = tn; }''')
rep('''  (void)my_d;''','''#ifdef OKKT_D_CYC
  if (marks && step == 0 && tid == 64) printf("DCYC row  wave (micro-step 6): barrier1 -> loads landed %lld, solve %lld, panels written %lld, barrier2 passed %lld, deferred outputs done %lld, next barrier1 passed %lld\\n", cy[1] - cy[0], cy[2] - cy[1], cy[3] - cy[2], cy[4] - cy[3], cy[6] - cy[4], cy[5] - cy[6]);
  if (marks && step == 0 && tid == 128) printf("DCYC mfma wave (micro-step 6): barrier1 -> rest of the update done %lld, barrier2 passed %lld, head: operands read %lld, MFMAs done %lld, copied out %lld, barrier1 passed %lld\\n", cy[1] - cy[0], cy[2] - cy[1], cy[3] - cy[2], cy[4] - cy[3], cy[5] - cy[4], cy[6] - cy[5]);
#endif
  (void)my_d;''')
open(p,'w').write(s)
