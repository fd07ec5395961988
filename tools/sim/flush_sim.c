// Counts the line-atomics the HexPlane scatter walk issues (csrc/hexplane.hip::foot2_add_t, statement by statement: two-entry
// footprint cache, no MRU swap, shift reuse on the miss path, everything flushed at the end of a segment) for a given walk order,
// so that alternative orders can be priced on the CPU before a kernel is written.   gcc -O2 -o flush_sim flush_sim.c
// stdin-free: flush_sim <keys.bin> <flags.bin> <n> <W> <row:0|1> <seg_len> <entries:1|2>
//   keys[i]  = nw texel index of point i's footprint IN WALK ORDER (int32), flags[i] = bit0 ne in range, bit1 sw in range (uint8)
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

typedef struct { int key, fl; } Ent;   // key < 0: empty

static long flush_ent(const Ent* e, int row) {
  if (e->key < 0) return 0;
  long n = 1;
  if (e->fl & 1) n++;
  if (!row && (e->fl & 2)) n++;
  if (!row && (e->fl & 3) == 3) n++;
  return n;
}

int main(int argc, char** argv) {
  if (argc < 8) return 2;
  long n = atol(argv[3]);
  int W = atoi(argv[4]), row = atoi(argv[5]), seg = atoi(argv[6]), entries = atoi(argv[7]);
  int32_t* keys = malloc(n * 4);
  uint8_t* flags = malloc(n);
  FILE* f = fopen(argv[1], "rb"); if (!f || fread(keys, 4, n, f) != (size_t)n) return 3; fclose(f);
  f = fopen(argv[2], "rb"); if (!f || fread(flags, 1, n, f) != (size_t)n) return 3; fclose(f);
  long atomics = 0, misses = 0, shifts = 0, seg_end = 0;
  for (long s0 = 0; s0 < n; s0 += seg) {
    long s1 = s0 + seg < n ? s0 + seg : n;
    Ent e[2] = {{-1, 0}, {-1, 0}};
    int mru = 0;
    for (long i = s0; i < s1; i++) {
      const int k = keys[i], fl = flags[i] & (row ? 1 : 3);
      int h0 = (k == e[0].key && fl == e[0].fl), h1 = entries == 2 && (k == e[1].key && fl == e[1].fl);
      if (!(h0 || h1)) {
        misses++;
        const int m1 = entries == 2 ? mru : 0;
        const Ent* m = &e[m1];
        const int down = !row && m->key >= 0 && k == m->key + W;
        const int right = m->key >= 0 && k == m->key + 1 && (m->fl & 1);
        const int shift = down || right;
        const int w1 = entries == 2 ? (shift ? m1 : !m1) : 0;
        const Ent* v = &e[w1];
        if (v->key >= 0) {
          atomics += 1;                                                  // nw
          if ((v->fl & 1) && !right) atomics++;
          if (!row && (v->fl & 2) && !down) atomics++;
          if (!row && (v->fl & 3) == 3 && !shift) atomics++;
          if (shift) shifts++;
        }
        e[w1].key = k; e[w1].fl = fl;
        h1 = w1; h0 = !w1;
      }
      mru = h1 ? 1 : 0;
    }
    long t = flush_ent(&e[0], row) + (entries == 2 ? flush_ent(&e[1], row) : 0);
    atomics += t; seg_end += t;
  }
  printf("%ld %ld %ld %ld\n", atomics, misses, shifts, seg_end);
  return 0;
}
