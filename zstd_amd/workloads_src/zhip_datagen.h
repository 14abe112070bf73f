// zhip_datagen.h — host-side synthetic input generator used by bench.py and the tests.
// Byte-compatible with the reference's programs/datagen.c (`datagen -g<size> -P<pct> -s<seed>` = RDG_genStdout,
// :155-186; and RDG_genBuffer, :144-153, which `zstd -b -P` uses): same xorshift-free LCG (:45-56), literal
// distribution table (:60-75) and match/literal state machine (:96-141).  tests/test_host_params.py compares both
// modes with the real thing.  No corpora (Silesia, enwik9) exist on the box, so this is the bench input.
#pragma once
#include <stdint.h>
#include <stddef.h>
#include <string.h>
#include <stdlib.h>

namespace zhip {

struct Rdg {
    uint32_t seed; uint8_t ldt[8192]; double matchProba;
    static uint32_t rnd(uint32_t* s) { uint32_t r = *s; r *= 2654435761U; r ^= 2246822519U; r = (r << 13) | (r >> 19); *s = r; return r >> 5; }
    static uint32_t rndLen(uint32_t* s) { if (rnd(s) & 7) return rnd(s) & 0xF; return (rnd(s) & 0x1FF) + 0xF; }
    void init(double mp, double lp, uint32_t sd)
    {
        seed = sd; matchProba = mp;
        memset(ldt, '0', sizeof(ldt));
        if (lp <= 0.0) lp = mp / 4.5;
        uint32_t const ld = (uint32_t)(lp * 256 + 0.001);
        uint8_t const first = ld ? '(' : 0, last = ld ? '}' : 255; uint8_t ch = ld ? '0' : 0;
        for (uint32_t u = 0; u < 8192; ) {
            uint32_t const w = (((8192 - u) * ld) >> 8) + 1;
            uint32_t const end = u + w < 8192 ? u + w : 8192;
            while (u < end) ldt[u++] = ch;
            ch++; if (ch > last) ch = first;
        }
    }
    // RDG_genBlock (:96-141): fills b[prefix .. size) using b[0 .. prefix) as history
    void block(uint8_t* b, size_t size, size_t prefix)
    {
        uint32_t const mp32 = (uint32_t)(32768 * matchProba);
        size_t pos = prefix; uint32_t prevOffset = 1;
        while (matchProba >= 1.0) {
            size_t size0 = rnd(&seed) & 3;
            size0 = (size_t)1 << (16 + size0 * 2);
            size0 += rnd(&seed) & (size0 - 1);
            if (size < pos + size0) { memset(b + pos, 0, size - pos); return; }
            memset(b + pos, 0, size0); pos += size0;
            b[pos - 1] = ldt[rnd(&seed) & 8191];
        }
        if (pos == 0) { b[0] = ldt[rnd(&seed) & 8191]; pos = 1; }
        while (pos < size) {
            if ((rnd(&seed) & 0x7FFF) < mp32) {
                uint32_t const length = rndLen(&seed) + 4;
                uint32_t const d = (uint32_t)(pos + length < size ? pos + length : size);
                uint32_t const repeatOffset = (rnd(&seed) & 15) == 2;
                uint32_t const randOffset = (rnd(&seed) & 0x7FFF) + 1;
                uint32_t const offset = repeatOffset ? prevOffset : (uint32_t)(randOffset < pos ? randOffset : pos);
                size_t match = pos - offset;
                while (pos < d) b[pos++] = b[match++];
                prevOffset = offset;
            } else {
                uint32_t const length = rndLen(&seed);
                uint32_t const d = (uint32_t)(pos + length < size ? pos + length : size);
                while (pos < d) b[pos++] = ldt[rnd(&seed) & 8191];
            }
        }
    }
};

// mode 0: RDG_genBuffer (one block over the whole buffer); mode 1: RDG_genStdout (32 KB dictionary + 128 KB blocks)
static inline void datagen(void* buffer, size_t size, double matchProba, double litProba, uint32_t seed, int mode)
{
    Rdg* g = (Rdg*)malloc(sizeof(Rdg));
    g->init(matchProba, litProba, seed);
    if (mode == 0) { if (size) g->block((uint8_t*)buffer, size, 0); free(g); return; }
    size_t const blk = 128 << 10, dict = 32 << 10;
    uint8_t* buff = (uint8_t*)malloc(dict + blk);
    g->block(buff, dict, 0);
    size_t total = 0;
    while (total < size) {
        size_t const gen = size - total < blk ? size - total : blk;
        g->block(buff, dict + blk, dict);
        memcpy((uint8_t*)buffer + total, buff, gen);
        total += gen;
        memcpy(buff, buff + blk, dict);
    }
    free(buff); free(g);
}

}  // namespace zhip
