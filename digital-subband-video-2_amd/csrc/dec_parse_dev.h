// dec_parse_dev.h -- the decoder's plane-section parse on the device (dec_parse_dev.hip).
#pragma once

#include "dev.h"

namespace dsv2 {

struct DecScanBases { // ScanGeom::base of a plane class: first scan position of {LL region, 3 levels x 3 subbands}, and the total
    int base[11];
};

// one plane section: the packet as staged in device memory (16-byte aligned, zero bytes behind it, >= 2 KB + 64 readable bytes
// behind its end), where the symbol codes start, where the section ends, and where the results go
struct DecParseJob {
    const uint8_t *pkt;
    uint32_t data_bitpos; // first bit of the first run (behind the 24-bit symbol count)
    uint32_t limit_bits;  // BitReader::limit of the host parse: (packet length + 8) * 8
    uint32_t end_byte;    // section start + plane length: the overrun check of hzcc.c:525-529
    int runs;             // symbol count from the header
    int cap;              // entries pos / val hold (min(runs, coefficients of the plane))
    int chroma;           // which DecScanBases applies
    uint32_t *pos;        // out: scan positions ...
    int32_t *val;         // ... and values, ascending position
    int *seg_out;         // out: DequantJob::seg of this plane ({LL, level 0, 1, 2} symbol counts; zeros for a damaged section)
    int *fail;            // out: 1 = damaged section (its residual plane is to stay zero)
};

void dec_parse_planes(hipStream_t s, const DecParseJob *d_jobs, int n, const DecScanBases &luma, const DecScanBases &chroma);
// zero fill of job k's dst (bytes, multiple of 16) iff d_flags[k] != 0
void zero_linear_if_batch(hipStream_t s, const CopyJob *d_jobs, const int *d_flags, int n, size_t max_bytes);

} // namespace dsv2
