// parser_fuzz.cpp -- the decoder's untrusted-input parser (csrc/dec_parse.h + csrc/entropy.cpp) on the CPU under AddressSanitizer
// and UndefinedBehaviorSanitizer.  No GPU, no device call: built and run by tests/test_parser_asan.py (and `make -C tests/..`
// by hand: see that file for the command line).
//
//   parser_fuzz <packets.bin> <iterations> <seed>
//   packets.bin: repeated { u32 length (little endian), bytes } -- the packets of one or more .dsv streams in order (metadata
//   packets included: they set the geometry the picture packets are parsed with)
//
// Every packet is first parsed as it is (a valid stream: every plane must decode), then `iterations` damaged copies of it:
// flipped bytes, overwritten length fields, truncations, spliced tails.  The process must neither crash nor trip a sanitizer;
// what the parser RETURNS for a damaged packet is not checked here (tests/test_gpu_robustness.py compares that with the
// reference decoder).
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "../digital-subband-video-2_amd/csrc/dec_parse.h"

using namespace dsv2;
using namespace dsv2::decparse;

namespace {

struct Rng {
    uint64_t s;
    uint32_t next()
    {
        s = s * 6364136223846793005ull + 1442695040888963407ull;
        return (uint32_t) (s >> 33);
    }
    uint32_t below(uint32_t n) { return n ? next() % n : 0; }
};

struct Stats {
    long parsed = 0, errors = 0, planes_ok = 0, planes_bad = 0, symbols = 0;
};

// one packet through the same steps as decoder.cpp: dec_parse (private copy with 64 zero bytes behind it, reads bounded by limit)
int parse_packet(DSV_DECODER *d, const uint8_t *data, unsigned len, Stats &st)
{
    std::vector<uint8_t> copy(data, data + len);
    copy.resize((size_t) len + 64, 0);
    const uint8_t *pkt = copy.data();
    BitReader br{pkt, 0};
    br.wide = true;
    br.limit = (len + 8) * 8;
    PictureHead hd;
    int rc = parse_head(br, d, hd);
    if (rc != kParsePicture) {
        st.errors += rc == DSV_DEC_ERROR;
        return rc;
    }
    const DSV_META *m = &d->vidmeta;
    const int nbh = (m->width + hd.blk_w - 1) / hd.blk_w, nbv = (m->height + hd.blk_h - 1) / hd.blk_h;
    int cw[3], ch[3];
    coef_dims(m->subsamp, m->width, m->height, cw, ch);
    ScanGeom scan[3];
    for (int c = 0; c < 3; c++) {
        make_scan(&scan[c], cw[c], ch[c]);
    }
    static SideBufs side;
    static std::vector<uint32_t> pos;
    static std::vector<int32_t> val;
    PictureBody body;
    // the heads-only form (what runs when the device parses the sections: dec_parse_dev.hip) must agree with the full parse on
    // every plane's verdict on the length field, its DC, and -- for sections it accepts -- place the section inside the limit
    {
        BitReader br2 = br;
        static SideBufs side2;
        PictureBody heads;
        parse_body(br2, pkt, hd.has_ref, nbh, nbv, scan, side2, pos, val, heads, true);
        BitReader br3 = br;
        parse_body(br3, pkt, hd.has_ref, nbh, nbv, scan, side, pos, val, body);
        for (int c = 0; c < 3; c++) {
            if ((heads.ok[c] < 0) != (body.ok[c] < 0) || (heads.ok[c] > 0 && heads.LL[c] != body.LL[c])) {
                fprintf(stderr, "heads-only parse disagrees with the full parse on plane %d\n", c);
                abort();
            }
            if (heads.ok[c] > 0 && (heads.head[c].data_bitpos > br.limit || heads.head[c].runs < 0 || heads.head[c].runs >= (1 << 24))) {
                fprintf(stderr, "heads-only parse: section %d outside the packet's limit\n", c);
                abort();
            }
        }
        br = br3;
    }
    st.parsed++;
    for (int c = 0; c < 3; c++) {
        (body.ok[c] > 0 ? st.planes_ok : st.planes_bad)++;
    }
    st.symbols += (long) body.nsym;
    // what the device phase would index with: positions must lie inside their plane's scan, sides inside the block grid
    size_t at = 0;
    for (int c = 0; c < 3; c++) {
        const size_t n = (size_t) (body.seg[c][0] + body.seg[c][1] + body.seg[c][2] + body.seg[c][3]);
        for (size_t k = 0; k < n; k++) {
            if (pos[at + k] >= (uint32_t) scan[c].base[10]) {
                fprintf(stderr, "symbol position %u outside the plane's scan (%d)\n", pos[at + k], scan[c].base[10]);
                abort();
            }
        }
        at += n;
    }
    return DSV_DEC_OK;
}

} // namespace

int main(int argc, char **argv)
{
    if (argc < 4) {
        fprintf(stderr, "usage: parser_fuzz packets.bin iterations seed\n");
        return 2;
    }
    FILE *f = fopen(argv[1], "rb");
    if (!f) {
        return 2;
    }
    std::vector<std::vector<uint8_t>> packets;
    for (;;) {
        uint32_t n;
        if (fread(&n, 4, 1, f) != 1) {
            break;
        }
        std::vector<uint8_t> p(n);
        if (n && fread(p.data(), 1, n, f) != n) {
            return 2;
        }
        packets.push_back(p);
    }
    fclose(f);
    const int iters = atoi(argv[2]);
    Rng rng{(uint64_t) atoll(argv[3]) * 2654435761ull + 1};
    DSV_DECODER dec;
    memset(&dec, 0, sizeof(dec));
    Stats clean, dirty;
    for (size_t k = 0; k < packets.size(); k++) {
        const std::vector<uint8_t> &p = packets[k];
        int rc = parse_packet(&dec, p.data(), (unsigned) p.size(), clean);
        if (rc == DSV_DEC_ERROR) {
            fprintf(stderr, "valid packet %zu was refused\n", k);
            return 1;
        }
        const DSV_META keep = dec.vidmeta; // (damaged metadata packets must not poison the geometry of the packets behind them)
        const int keep_got = dec.got_metadata;
        for (int it = 0; it < iters; it++) {
            std::vector<uint8_t> q = p;
            const int kind = (int) rng.below(6);
            const int nmut = 1 + (int) rng.below(4);
            for (int m = 0; m < nmut && !q.empty(); m++) {
                const uint32_t i = rng.below((uint32_t) q.size());
                switch (kind) {
                    case 0: q[i] ^= (uint8_t) (1 + rng.below(255)); break;                       // flipped bits anywhere
                    case 1: q[i] = 0xff; break;                                                   // long unary runs
                    case 2:                                                                       // a 32-bit field set to a huge / tiny value
                        for (int b = 0; b < 4 && i + b < q.size(); b++) {
                            q[i + b] = (uint8_t) (rng.below(2) ? 0xff : 0x00);
                        }
                        break;
                    case 3: q.resize(i); break;                                                   // truncated
                    case 4:                                                                       // tail replaced by noise
                        for (size_t j = i; j < q.size(); j++) {
                            q[j] = (uint8_t) rng.next();
                        }
                        break;
                    default:                                                                      // damage in the head: side information
                        if (q.size() > 24) {
                            q[14 + rng.below((uint32_t) (q.size() < 200 ? q.size() - 14 : 186))] ^= (uint8_t) (1 + rng.below(255));
                        }
                        break;
                }
            }
            parse_packet(&dec, q.data(), (unsigned) q.size(), dirty);
            dec.vidmeta = keep;
            dec.got_metadata = keep_got;
        }
    }
    printf("{\"packets\": %zu, \"clean_pictures\": %ld, \"clean_planes_ok\": %ld, \"clean_planes_bad\": %ld, \"damaged_parsed\": %ld, "
           "\"damaged_refused\": %ld, \"damaged_planes_bad\": %ld, \"damaged_symbols\": %ld}\n",
           packets.size(), clean.parsed, clean.planes_ok, clean.planes_bad, dirty.parsed, dirty.errors, dirty.planes_bad, dirty.symbols);
    return clean.planes_bad ? 1 : 0;
}
