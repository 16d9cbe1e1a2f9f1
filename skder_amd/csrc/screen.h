// screen.h -- marker screen entry point
#pragma once
#include "engine.h"
// rows: genome indices of `queries` to screen against all genomes of `refs`.
// triangle: refs == queries and only partners j > row are considered.
// Output: ordered pair list (row, partner).
void screen_pairs(skder_sketches *refs, skder_sketches *queries, const std::vector<uint32_t> &rows, bool triangle,
                  double screen_pct, std::vector<uint32_t> &pair_row, std::vector<uint32_t> &pair_partner);
