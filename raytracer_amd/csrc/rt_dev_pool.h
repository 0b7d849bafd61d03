// rt_dev_pool.h -- pooled traversal: the same closest hit as trace_run (rt_dev_scene.h), scheduled differently.
//
// k_traverse gives every lane one ray and walks it to the end: a wave pays every step that any of its 64 rays takes, and
// with incoherent rays in a two-level scene (the 52,743-instance field) 70 % of the lanes of an issued vector instruction
// are idle (PMC, profiles/pmc_traffic.json: SQ_THREAD_CYCLES_VALU / SQ_INSTS_VALU / 64 = 0.29) while the vector ALUs
// are busy 90 % of the time -- the kernel is bound by instructions that mostly do nothing.
//
// Here a workgroup keeps a POOL of POOL_RAYS rays whose whole traversal state lives in LDS (stack rows and ~30 words per
// ray), and works in rounds.  Every round sorts the rays by what they need next --
//     NODE  a step down the tree          LEAF  the triangles of a bottom-level leaf
//     INST  entering an instance          FREE  the slot is done: write its hit, take the next ray of the queue
// -- into lists (ballots and a scan across the waves), cuts the lists into batches of 64, and each wave takes one batch:
// it loads the state its phase needs from LDS, runs the phase with (nearly) all lanes busy, and stores what changed.
// A ray moves from wave to wave and lane to lane as it goes; the closest hit does not depend on the order of the walk
// (S5: ties go to the lowest id), so hits, t, u, v are the same bits as k_traverse's.  Workgroups are persistent: each
// takes queue segments from a global counter until none are left, so the pools stay full until the launch ends.
#pragma once
#include "rt_dev_scene.h"

namespace rtd {

static constexpr int POOL_RAYS = 256;    // ray slots = threads of a workgroup
enum PoolField { PF_CUR, PF_SP, PF_IDX, PF_IDY, PF_IDZ, PF_CNX, PF_CNY, PF_CNZ, PF_CFX, PF_CFY, PF_CFZ,   // the current level's BoxRay
                 PF_OX, PF_OY, PF_OZ, PF_SX, PF_SY, PF_SZ, PF_KZ,                                        // RayPrep (kz | flip | bottom << 8)
                 PF_NLO, PF_NHI, PF_TLO, PF_THI,                                                         // node / triangle arrays of the current level
                 PF_T, PF_UN, PF_VN, PF_DET, PF_INST, PF_TRI, PF_PRIM,                                   // ClosestHit
                 PF_QI, PF_CINST, PF_CLS, PF_COUNT };
enum PoolClass { PC_NODE = 0, PC_LEAF = 1, PC_INST = 2, PC_FREE = 3, PC_IDLE = 4 };

// stack of one pooled ray: rows of POOL_RAYS words, row 0 the RT_TRAVERSAL_DONE sentinel (the interface node_step expects)
struct PoolStack {
    RT_LDS int* lds; // &stack[slot]
    int sp;
    RT_DEV int top() const { return lds[sp * POOL_RAYS]; }
    RT_DEV void put(int v, bool) { lds[(sp + 1) * POOL_RAYS] = v; }
    RT_DEV void push(int v) { put(v, true); sp++; }
    RT_DEV int pop() { const int v = top(); sp--; return v; }
};

RT_DEV int pool_class_of(int cur, bool bottom) {
    if (cur >= 0) return PC_NODE;
    if (cur == RT_TRAVERSAL_DONE) return PC_FREE;
    return bottom ? PC_LEAF : PC_INST; // (the marker never gets here: pool_leave_instance has popped it)
}

} // namespace rtd
