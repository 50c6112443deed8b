#pragma once
#include <cstdint>
#include <string>
#include <vector>
#include "sicelore_mi.h"
namespace smi {
void set_error(const std::string &msg);
uint32_t host_crc32(uint32_t crc, const uint8_t *p, size_t n);
int host_inflate_exact(const uint8_t *in, size_t n_in, uint8_t *out, size_t n_out);
int host_gunzip(const uint8_t *in, size_t n_in, size_t *in_pos, uint8_t *out, size_t cap, size_t *out_pos);
}
