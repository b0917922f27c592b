// Static assignment of the lower-triangular 16x16 tiles of an (16 NT) x (16 NT) symmetric product to the 4
// waves of a workgroup: wave W owns the tile ROWS RowsOf<NT, W>::RA and ::RB (-1 = none), i.e. tiles
// (RA, 0..RA) and (RB, 0..RB).  Rows are paired (NT-1-w, w-1)-style so every wave owns about NT+1 tiles.
#pragma once

template <int NT, int W> struct RowsOf;
#define DKT_ROWS(NT_, W_, RA_, RB_) \
    template <> struct RowsOf<NT_, W_> { static constexpr int RA = RA_, RB = RB_; };
DKT_ROWS(1, 0, 0, -1) DKT_ROWS(1, 1, -1, -1) DKT_ROWS(1, 2, -1, -1) DKT_ROWS(1, 3, -1, -1)
DKT_ROWS(2, 0, 1, -1) DKT_ROWS(2, 1, 0, -1) DKT_ROWS(2, 2, -1, -1) DKT_ROWS(2, 3, -1, -1)
DKT_ROWS(3, 0, 2, -1) DKT_ROWS(3, 1, 1, 0) DKT_ROWS(3, 2, -1, -1) DKT_ROWS(3, 3, -1, -1)
DKT_ROWS(4, 0, 3, -1) DKT_ROWS(4, 1, 2, -1) DKT_ROWS(4, 2, 1, 0) DKT_ROWS(4, 3, -1, -1)
DKT_ROWS(5, 0, 4, -1) DKT_ROWS(5, 1, 3, 0) DKT_ROWS(5, 2, 2, 1) DKT_ROWS(5, 3, -1, -1)
DKT_ROWS(6, 0, 5, -1) DKT_ROWS(6, 1, 4, 0) DKT_ROWS(6, 2, 3, 1) DKT_ROWS(6, 3, 2, -1)
DKT_ROWS(7, 0, 6, -1) DKT_ROWS(7, 1, 5, 0) DKT_ROWS(7, 2, 4, 1) DKT_ROWS(7, 3, 3, 2)
DKT_ROWS(8, 0, 7, 0) DKT_ROWS(8, 1, 6, 1) DKT_ROWS(8, 2, 5, 2) DKT_ROWS(8, 3, 4, 3)
#undef DKT_ROWS
