// entropy_gpu.h -- plane sections of the picture packet assembled on the GPU (entropy_gpu.hip)
#pragma once

#include "quant.h"

namespace dsv2 {

constexpr int kEntChunk = 1024; // symbols per chunk of the state-transfer decomposition

struct EntGeom { // what locates a symbol: plane boundaries in the stream's symbol positions, subband boundaries per plane
    int qv_off[4];
    int base[3][11];
};
EntGeom ent_geom(const size_t qv_off[4], const ScanGeom scan[3]);

struct EntJob { // one stream of a lockstep step (device table entry)
    const uint32_t *pos; // compacted symbols, ascending scan position over the three planes
    const int32_t *val;
    const int *total;    // their number (device)
    int list_cap;        // ... of which pos / val hold at most this many: a picture with more raises ENT_LIST_OVERFLOW and is left alone
    const int32_t *ll;   // the three DC coefficients, sent raw (device)
    uint16_t *tables, *chunk_vk;
    uint32_t *chunk_bits, *chunk_off;
    uint2 *chunk_join;   // per chunk: {m | q0 << 16 where its 256 trajectories had joined (m, m + 1), or all ones; the pair's end states}
    uint8_t *ksym;       // per symbol, chunk-major: its threshold >> 3, then the state >> 3 it meets (round 3 form: its parameter, list order)
    uint8_t *out;
    uint32_t out_cap;
    int *info;
    uint8_t *host_out; // pinned: the finished bytes (three plane sections back to back) ...
    int *host_info;    // ... and the 16 info words: [0] flags, [5..7] bytes of each section, [8] total bytes
    uint32_t host_cap;
};
// flags: 1 adaptive state left the tabulated range, 2 output buffer too small, 4 absurd code length (all three: the host
// codes the picture from the symbol list instead), 8 finished but larger than the pinned mirror (fetch `out` by copy)
// 16: the picture has more symbols than the stream's compaction lists hold -- nothing was coded; the encoder grows the lists and
// has the picture's symbols worked out again (encoder.cpp: redo_overflow)
enum { ENT_FALLBACK_MASK = 7, ENT_NOT_MIRRORED = 8, ENT_LIST_OVERFLOW = 16, ENT_INFO_PBYTES = 5, ENT_INFO_TOTAL = 8 };

struct EntBuffers { // per encoder instance
    uint16_t *tables = nullptr, *chunk_vk = nullptr;
    uint32_t *chunk_bits = nullptr, *chunk_off = nullptr;
    void *chunk_join = nullptr;
    uint8_t *ksym = nullptr, *out = nullptr;
    int *info = nullptr;
    uint8_t *host_out = nullptr;
    int *host_info = nullptr;
    uint32_t out_cap = 0, host_cap = 0;
    void ensure(size_t nsym_cap, uint32_t out_bytes, uint32_t host_bytes);
    void release();
    EntJob job(const uint32_t *pos, const int32_t *val, const int *total, const int32_t *ll, size_t list_cap) const;
    size_t nsym_cap = 0; // symbols the chunk tables / parameter bytes are sized for
};

// n streams of identical geometry; chunk_slots = workgroups per (stream, plane) that share that plane's chunks
void entropy_gpu_jobs(hipStream_t s, const EntJob *d_jobs, int n, const EntGeom &g, int chunk_slots);

} // namespace dsv2
