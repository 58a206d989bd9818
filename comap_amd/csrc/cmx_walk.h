// The tree walk of one rate-class pass of the mapping kernel, written ONCE and instantiated three times:
//   * host, Recorder backend  (cmx_host_model.cpp): lists the operators and workspace loads in program order -- the
//     operator stream and the load schedule the device follows;
//   * host, Numeric backend   (cmx_host_model.cpp): runs the pass in plain doubles from the device layouts, consuming
//     that stream exactly as the device does, and is compared with a direct pruning computation before a context is
//     accepted (verify_walk);
//   * device backend          (cmx_kernels.hip: map_sites_wave): registers are S-vectors of the wave's 64 sites.
// So the control flow that decides WHICH operator every product / leaf gather applies exists in one place.
//
// Algorithm ("message scheme").  M_n = P_n D_n is the message a node sends up its branch, D_n the product
// of its children's messages.  Inside pass (post-order over the visited nodes): D_n in R0, M_n -> R1 and the workspace.
// Outside pass (reverse order), node f with U_f = the message arriving at the top of its branch in R0:
//     W = J_f^T U_f,  count_f = sum W o M_a o M_b,  Up = P_f^T U_f,  U_a = Up o M_b,  U_b = Up o M_a
// i.e. two matrix products per internal branch here and one in the inside pass; sibling messages are loaded, never
// recomputed.  "Inlined cherries" (an internal node with two leaf children under a binary parent) are never visited:
// their message is rebuilt from two leaf gathers and one product where needed and their three branches are counted
// where their outside message is produced.  A node's LAST visited child hands its message over in registers (R1 on the
// way up, R0 on the way down); everything else goes through the per-wave workspace, written once.
//
// Registers: R0..R3.  Only R1 (inside) / R0 (outside) are live from one node to the next; the backends' kill<R>() tells
// a compiler so.  Templates name registers at compile time; the device maps them to VGPR arrays.
#pragma once

#ifdef __HIPCC__
#define CMX_HD __host__ __device__ __forceinline__
#else
#define CMX_HD inline
#endif

namespace cmx {

// per-visited-node record, 16 ints.  The walk only knows BINARY nodes: the host splits every multifurcation (the
// trifurcating root of an unrooted tree included) into a chain of pseudo nodes joined by zero-length branches -- no
// operator, no count, M = D and Up = U on them -- so that no loop over children exists on the device.
enum { REC_NODE = 0 /* branch / node id, -1 for a pseudo node */, REC_SLOT = 1 /* workspace slot */, REC_FLAGS = 3,
       REC_A = 4 /* 5 ints */, REC_B = 9 /* 5 ints */ };
enum { CH_KIND = 0, CH_NODE = 1, CH_SLOT = 2, CH_L1 = 3, CH_L2 = 4 };   // child descriptor (L1 / L2: leaf nodes of an inlined cherry)
enum { KIND_LEAF = 0, KIND_STORED = 1, KIND_CHERRY = 2 };
enum { FLAG_PSEUDO = 1,       // zero-length branch of a split multifurcation
       FLAG_HAND = 2,         // child B is the node visited right before (inside) / right after (outside) this one
       FLAG_U_HANDED = 4,     // this node's outside message arrives in R0 (its parent is the node visited right before it)
       FLAG_ROOT = 8 };
enum { WS_M = 0, WS_U = 1 };  // workspace arrays: messages of the inside pass, outside messages
enum { OPER_P = -1 };         // which operator of a branch: OPER_P = transition matrix, k >= 0 = count operator k

// Cherry tables (round 4; class-fused nucleotide models on fully resolved alignments, i.e. the null's): with four states a
// cherry's two symbols take 16 values, so everything an inlined cherry contributes is a row of a 16-row table indexed by
// (symbol of l1, symbol of l2) -- its message M_c = P_c (M_l1 o M_l2) (table 0), and, as dot products with its outside
// message U_c, the counts of its own branch (table 1 + k: J_c^k (M_l1 o M_l2)) and of its two leaf branches (tables
// 1 + K + k and 1 + 2 K + k: P_c (J_l1^k o M_l2), P_c (M_l1 o J_l2^k)).  A backend with kCherryTables = true gets
// cset / cdot instead of the 3 operator ops of a cherry's message and the 5 + 3 K of its outside visit: one gather each.
//
// Backend concept (all members force-inlined on the device):
//   static constexpr bool kCherryTables;  cset<D>(node, l1, l2)  [D = table 0 row];  cdot<S>(node, l1, l2, table, row)  [count]
//   rec(v, int (&r)[16])
//   lset<D>(leaf, which); lmul<S, D>(leaf, which)  [D = S o row];  ldot<S>(leaf, which, row)  [count]
//   mv<S, D, TR>(node, which)  [D = M S or M^T S];  load<D>(arr, slot); store<S>(arr, slot)
//   mov<D, S>(); mul<D, S>() [D *= S]; mulup() [R1 *= R3, R2 *= R3]; setpi<D>(); rootl<S>(); dot3(row); kill<R>()
template <class BE>
CMX_HD void walk_cherry_message(BE& be, int node, int l1, int l2) {   // M of an inlined cherry -> R1, through R3
  if constexpr (BE::kCherryTables) {
    be.template kill<1>();
    be.template cset<1>(node, l1, l2);
    return;
  }
  be.template kill<3>();
  be.template lset<3>(l1, OPER_P);
  be.template lmul<3, 3>(l2, OPER_P);
  be.template kill<1>();
  be.template mv<3, 1, false>(node, OPER_P);
}

// binary node, outside pass: message of child SIDE (0 = A -> R1, 1 = B -> R2).  SIDE is a template parameter on purpose:
// a run-time loop over the two children would make every register a loop-carried value of that loop (the register
// allocator then shuffles all four vectors around its back edge).
template <int SIDE, class BE>
CMX_HD void walk_child_message(BE& be, const int (&r)[16]) {
  constexpr int o = SIDE ? REC_B : REC_A;
  const int kind = r[o + CH_KIND];
  if (kind == KIND_LEAF) {
    be.template lset<SIDE ? 2 : 1>(r[o + CH_NODE], OPER_P);
  } else if (kind == KIND_STORED) {
    be.template load<SIDE ? 2 : 1>(WS_M, r[o + CH_SLOT]);
  } else {
    walk_cherry_message(be, r[o + CH_NODE], r[o + CH_L1], r[o + CH_L2]);
    if (SIDE) be.template mov<2, 1>();
  }
}

// binary node, outside pass: what happens to the outside message of child SIDE (U_a in R2, U_b in R1)
template <int SIDE, class BE>
CMX_HD void walk_child_dispose(BE& be, const int (&r)[16], int K) {
  constexpr int o = SIDE ? REC_B : REC_A;
  constexpr int UR = SIDE ? 1 : 2;
  const int kind = r[o + CH_KIND], node = r[o + CH_NODE];
  if (kind == KIND_LEAF) {
    for (int k = 0; k < K; ++k) be.template ldot<UR>(node, k, node * K + k);
  } else if (kind == KIND_STORED && !(SIDE && (r[REC_FLAGS] & FLAG_HAND))) {
    be.template store<UR>(WS_U, r[o + CH_SLOT]);
  } else {
    if constexpr (BE::kCherryTables) {
      if (kind == KIND_CHERRY) {
        // the cherry's whole outside visit: three dot products of its outside message with table rows per substitution type
        const int l1 = r[o + CH_L1], l2 = r[o + CH_L2];
        for (int k = 0; k < K; ++k) {
          be.template cdot<UR>(node, l1, l2, 1 + k, node * K + k);
          be.template cdot<UR>(node, l1, l2, 1 + K + k, l1 * K + k);
          be.template cdot<UR>(node, l1, l2, 1 + 2 * K + k, l2 * K + k);
        }
        return;
      }
    }
    // cherry: its visit happens here, with U in R0; handed-over child: U stays in R0 for the next node
    be.template mov<0, UR>();
    if (kind == KIND_CHERRY) {
      const int l1 = r[o + CH_L1], l2 = r[o + CH_L2];
      for (int k = 0; k <= K; ++k) {
        be.template kill<3>();
        be.template mv<0, 3, true>(node, k < K ? k : OPER_P);      // W = J_c^T U_c, at last Up_c = P_c^T U_c
        if (k < K) {
          be.template lmul<3, 3>(l1, OPER_P);                      // W o M_l1
          be.template ldot<3>(l2, OPER_P, node * K + k);           // count_c = sum W o M_l1 o M_l2
        }
      }
      be.template kill<0>();
      be.template lmul<3, 0>(l2, OPER_P);                          // U_l1 = Up_c o M_l2
      for (int k = 0; k < K; ++k) be.template ldot<0>(l1, k, l1 * K + k);
      be.template lmul<3, 0>(l1, OPER_P);                          // U_l2 = Up_c o M_l1
      for (int k = 0; k < K; ++k) be.template ldot<0>(l2, k, l2 * K + k);
    }
  }
}

template <class BE>
CMX_HD void walk_pass(BE& be, int NV, int K) {
  // ------------------------------------------------------------------ inside pass
  for (int v = 0; v < NV; ++v) {
    int r[16];
    be.rec(v, r);
    const int flags = r[REC_FLAGS];
    be.template kill<0>();
    be.template kill<2>();
    be.template kill<3>();
    const int ka = r[REC_A + CH_KIND], kb = r[REC_B + CH_KIND];
    const bool hand = (flags & FLAG_HAND) != 0;            // M_b is in R1
    if (!hand) be.template kill<1>();
    if (hand && ka == KIND_LEAF) {
      be.template lmul<1, 0>(r[REC_A + CH_NODE], OPER_P);
    } else {
      if (ka == KIND_LEAF) {
        be.template lset<0>(r[REC_A + CH_NODE], OPER_P);
      } else if (ka == KIND_STORED) {
        be.template load<0>(WS_M, r[REC_A + CH_SLOT]);
      } else if constexpr (BE::kCherryTables) {
        be.template cset<0>(r[REC_A + CH_NODE], r[REC_A + CH_L1], r[REC_A + CH_L2]);
      } else {
        be.template lset<3>(r[REC_A + CH_L1], OPER_P);
        be.template lmul<3, 3>(r[REC_A + CH_L2], OPER_P);
        be.template mv<3, 0, false>(r[REC_A + CH_NODE], OPER_P);
      }
      if (hand) be.template mul<0, 1>();
    }
    if (!hand) {
      if (kb == KIND_LEAF) {
        be.template lmul<0, 0>(r[REC_B + CH_NODE], OPER_P);
      } else {
        if (kb == KIND_STORED) be.template load<1>(WS_M, r[REC_B + CH_SLOT]);
        else walk_cherry_message(be, r[REC_B + CH_NODE], r[REC_B + CH_L1], r[REC_B + CH_L2]);
        be.template mul<0, 1>();
      }
    }
    if (flags & FLAG_ROOT) {
      be.template rootl<0>();
    } else {
      be.template kill<1>();
      if (flags & FLAG_PSEUDO) be.template mov<1, 0>();                 // zero-length branch: M = D
      else be.template mv<0, 1, false>(r[REC_NODE], OPER_P);
      be.template store<1>(WS_M, r[REC_SLOT]);
    }
  }
  // ------------------------------------------------------------------ outside pass + joint counts
  for (int v = NV - 1; v >= 0; --v) {
    int r[16];
    be.rec(v, r);
    const int flags = r[REC_FLAGS], f = r[REC_NODE];
    be.template kill<1>();
    be.template kill<2>();
    be.template kill<3>();
    if (!(flags & FLAG_U_HANDED)) {
      be.template kill<0>();
      if (!(flags & FLAG_ROOT)) be.template load<0>(WS_U, r[REC_SLOT]);
    }
    // ---- messages of the children: M_b -> R2, M_a -> R1 (stored ones: the device only ISSUES the loads here; they are
    // first read by the count below, after the first product)
    walk_child_message<1>(be, r);
    walk_child_message<0>(be, r);
    // ---- count of branch f and its outside message below the branch: W = J^T U (K of them), Up = P^T U
    if (flags & FLAG_ROOT) {
      be.template setpi<3>();
    } else if (flags & FLAG_PSEUDO) {
      be.template mov<3, 0>();                                           // zero-length branch: Up = U, nothing to count
    } else {
      for (int k = 0; k <= K; ++k) {
        be.template kill<3>();
        be.template mv<0, 3, true>(f, k < K ? k : OPER_P);
        if (k < K) be.dot3(f * K + k);
      }
    }
    be.template kill<0>();
    // ---- outside messages of the children
    be.mulup();                                     // R1 = U_b, R2 = U_a
    walk_child_dispose<0>(be, r, K);
    walk_child_dispose<1>(be, r, K);
  }
}

}  // namespace cmx
