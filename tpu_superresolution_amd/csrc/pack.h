// Descriptor-driven weight packing (fp32 state_dict layout -> padded bf16 kernel layout) and its
// inverse for gradients (fp32 staging gradient in packed layout -> fp32 state_dict layout).
#pragma once
#include "common.h"

enum { PK_LINEAR = 0, PK_CONV = 1, PK_VEC = 2, PK_RPB = 3 };
enum { NM_DIRECT = 0, NM_QKV = 1, NM_PS = 2 };      // packed row index n -> source output-feature index
enum { KM_DIRECT = 0, KM_HEADS = 1 };               // packed col index k -> source input-feature index

// One packing job.  The packed (non-transposed) matrix is [NP][KP] (conv: KP = 9*CinP, tap-major).
//   PK_LINEAR: src[o][i], o = nmap(n) < Nreal, i = kmap(k) < Kreal
//   PK_CONV:   src[co][ci][tap] ([Cout][Cin][3][3]); n -> co via nmap; k = tap*CinP + ci
//   PK_VEC:    fp32 vector [NP]: dstf[n] = src[nmap(n)]               (biases; dst is the fp32 side buffer)
//   PK_RPB:    dense bias fp32 [nH][64][64] from table [225][nH]      (dst is the fp32 side buffer)
// transpose (LINEAR/CONV only, pack only): write dst[k'][n] instead (dgrad weights).  For CONV the
// transposed form is [CinP][9][NP] with the tap flipped (tap' = 8 - tap): dgrad == conv with it.
struct PackDesc {
  int kind, transpose;
  long long src;     // offset (floats) into the flat parameter buffer
  long long dst;     // offset into the packed bf16 buffer (elements) or the fp32 side buffer (floats)
  int NP, KP;
  int Nreal, Kreal;  // LINEAR: rows/cols of src; CONV: Cout/Cin
  int nmap, kmap;
  int nH, dh, CA;    // head geometry (NM_QKV / KM_HEADS): CA = nH*32
  int r, Cs;         // NM_PS: n = ij*Cs + c  <->  co = c*r*r + ij
  int CinP;          // CONV
  int blk0;          // first workgroup of this job (prefix sum), filled by the host
};

int srk_launch_pack(const PackDesc* d_descs, int ndesc, int total_blocks, const float* params, bf16_t* packed,
                    float* side, hipStream_t stream);
int srk_launch_unpack_grads(const PackDesc* d_descs, int ndesc, int total_blocks, const float* gstage_w,
                            const float* gstage_side, float* grads, hipStream_t stream);
