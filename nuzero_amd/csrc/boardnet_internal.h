// Library-internal surface of the board nets (boardnet.hip) for the persistent SCS self-play kernel (scs_search.hip):
// the layer program of a network evaluated by ONE wavefront for ONE position, activations in the wavefront's own block
// of LDS, weights streamed from L2.  Not part of the C ABI.
#pragma once
#include <string>

#include "fused16_dev.hpp"

struct nz_boardnet;

namespace nz {

struct WaveNet {
  const Fused16Program* prog;     // device memory; offsets are floats from the wavefront's LDS block
  int32_t lds_floats;             // size of that block
  int32_t stage_off, stage_floats;  // float32 image [hw][inp] the leaf's planes are written to before they are split into
                                  // the input pieces (it lies over trunk buffers: zeroed again once the pieces are made)
  int32_t inp, in_channels;       // padded / real input planes
  int32_t hw, rows, cols, planes, hex, n_ops;
  int64_t flops;                  // algorithmic float32 FLOPs per position
  int64_t mfmas;                  // v_mfma_f32_16x16x32_bf16 issued per position (two row tiles, six split terms, padding included)
};

// false (with the reason): this network has no per-wavefront form (architecture, board size, widths, LDS)
bool boardnet_wave_program(nz_boardnet* h, WaveNet* out, std::string* why);

}  // namespace nz
