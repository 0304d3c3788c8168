// The launches of a Conformer block issued from ONE C call (round 4).
//
// The training engine (mindaudio_amd/train/engine.py) used to walk a block's ~26 launches from Python: a wrapper call, a handful of
// tensor allocations and a ctypes call per launch, 4.2 ms of host time for the 8.7 ms step of cfg 4 - fine for one rank, not for eight
// ranks sharing one host's cores.  A block table holds, per (direction, block), the list of C-ABI calls the engine made for that block
// in one step of a given batch shape - entry point + argument words + copies of the host structs the arguments point to - and
// ma_conformer_block_fwd_train / ma_conformer_block_bwd_train re-issue that list with the step's dropout seed and stream.  The buffers
// named by the entries stay allocated for as long as the table lives (the engine keeps them referenced), so a replayed block reads and
// writes exactly the addresses the walked block did: same kernels, same arguments, same order - bit-identical results by construction
// (tests/test_train_step_gpu.py compares the masters).  What the calls compute is the training step of
// /root/reference/mindaudio/utils/train_one_step.py:13-48 over models/conformer.py:109-156; the table is plumbing, not arithmetic.
//
// The typed calls are generated from include/mindaudio_amd.h (tools/gen_block_table.py -> block_table_calls.inc), so an entry point
// whose signature changes cannot be replayed with stale argument types.
#include <cstring>
#include <new>
#include <vector>

#include "launch.h"

namespace {

union Word {
  void* p;
  int64_t i;
  double d;
};

struct Call {
  int32_t fn;
  int32_t n_words;
  size_t word0;  // index of the first word in Segment::words
  size_t blob0;  // byte offset of the call's blob in Segment::blob
  size_t blob_bytes;
};

struct Segment {
  std::vector<Call> calls;
  std::vector<Word> words;
  std::vector<unsigned char> blob;  // 8-byte aligned pieces (host structs)
};

struct Name {
  const char* name;
  int32_t n_params;
  uint64_t seeds;  // bit k: parameter k is a dropout seed (replaced by the replaying call's)
};
const Name kNames[] = {
#define MA_BLOCK_TABLE_NAMES
#include "block_table_calls.inc"
#undef MA_BLOCK_TABLE_NAMES
};
constexpr int32_t kNumNames = (int32_t)(sizeof(kNames) / sizeof(kNames[0]));
const char* const kSkipped[] = {
#define MA_BLOCK_TABLE_SKIPPED
#include "block_table_calls.inc"
#undef MA_BLOCK_TABLE_SKIPPED
    nullptr};

struct Ctx {
  unsigned char* blob;  // the call's blob inside the segment
  size_t blob_bytes;
};

inline const void* host_blob(const Ctx& c, int64_t off) {
  return (off < 0 || (size_t)off >= c.blob_bytes) ? nullptr : (const void*)(c.blob + off);
}
// a ma_train_epilogue_t of the entry, carrying the replaying step's dropout seed
inline const void* host_epilogue(const Ctx& c, int64_t off, uint32_t seed) {
  if (off < 0 || (size_t)off + sizeof(ma_train_epilogue_t) > c.blob_bytes) return nullptr;
  ma_train_epilogue_t* e = reinterpret_cast<ma_train_epilogue_t*>(c.blob + off);
  e->seed = seed;
  return e;
}

int issue(int32_t fn, const Word* w, const Ctx& c, uint32_t s, ma_stream_t st) {
  switch (fn) {
#include "block_table_calls.inc"
    default:
      return MA_ERR_INVALID_ARG;
  }
}

constexpr int kMaxBlocks = 64;

}  // namespace

struct ma_block_table {
  Segment seg[2][kMaxBlocks];
  int32_t failed_call = -1;
};

extern "C" {

ma_block_table_t* ma_block_table_create(void) { return new (std::nothrow) ma_block_table(); }

void ma_block_table_destroy(ma_block_table_t* t) { delete t; }

int32_t ma_block_table_entry_point(const char* name) {
  if (name == nullptr) return -1;
  for (int32_t k = 0; k < kNumNames; ++k)
    if (std::strcmp(kNames[k].name, name) == 0) return k;
  for (const char* const* q = kSkipped; *q != nullptr; ++q)
    if (std::strcmp(*q, name) == 0) return -2;  // a launch, but not one the table can re-issue
  return -1;
}

int32_t ma_block_table_entry_point_params(int32_t fn) { return (fn < 0 || fn >= kNumNames) ? -1 : kNames[fn].n_params; }

uint64_t ma_block_table_entry_point_seeds(int32_t fn) { return (fn < 0 || fn >= kNumNames) ? 0 : kNames[fn].seeds; }

int ma_block_table_add(ma_block_table_t* t, int32_t backward, int32_t block, int32_t fn, const int64_t* words, int32_t n_words,
                       const void* blob, int64_t blob_bytes) {
  if (t == nullptr || block < 0 || block >= kMaxBlocks || fn < 0 || fn >= kNumNames || words == nullptr ||
      n_words != kNames[fn].n_params || blob_bytes < 0 || (blob_bytes > 0 && blob == nullptr) || (blob_bytes & 7))
    return MA_ERR_INVALID_ARG;
  Segment& s = t->seg[backward ? 1 : 0][block];
  Call c;
  c.fn = fn;
  c.n_words = n_words;
  c.word0 = s.words.size();
  c.blob0 = s.blob.size();
  c.blob_bytes = (size_t)blob_bytes;
  for (int32_t k = 0; k < n_words; ++k) {
    Word w;
    w.i = words[k];
    s.words.push_back(w);
  }
  if (blob_bytes > 0) {
    const unsigned char* b = static_cast<const unsigned char*>(blob);
    s.blob.insert(s.blob.end(), b, b + blob_bytes);
  }
  s.calls.push_back(c);
  return MA_OK;
}

int32_t ma_block_table_calls(const ma_block_table_t* t, int32_t backward, int32_t block) {
  if (t == nullptr || block < 0 || block >= kMaxBlocks) return -1;
  return (int32_t)t->seg[backward ? 1 : 0][block].calls.size();
}

int32_t ma_block_table_failed_call(const ma_block_table_t* t) { return t == nullptr ? -1 : t->failed_call; }

// what entry `call` of (direction, block) holds (the host side's tests read a table back through these)
int32_t ma_block_table_call_entry_point(const ma_block_table_t* t, int32_t backward, int32_t block, int32_t call) {
  if (t == nullptr || block < 0 || block >= kMaxBlocks) return -1;
  const Segment& s = t->seg[backward ? 1 : 0][block];
  return (call < 0 || (size_t)call >= s.calls.size()) ? -1 : s.calls[call].fn;
}

int64_t ma_block_table_call_word(const ma_block_table_t* t, int32_t backward, int32_t block, int32_t call, int32_t k) {
  if (t == nullptr || block < 0 || block >= kMaxBlocks) return 0;
  const Segment& s = t->seg[backward ? 1 : 0][block];
  if (call < 0 || (size_t)call >= s.calls.size() || k < 0 || k >= s.calls[call].n_words) return 0;
  return s.words[s.calls[call].word0 + k].i;
}

int64_t ma_block_table_call_blob(const ma_block_table_t* t, int32_t backward, int32_t block, int32_t call, void* out, int64_t bytes) {
  if (t == nullptr || block < 0 || block >= kMaxBlocks) return -1;
  const Segment& s = t->seg[backward ? 1 : 0][block];
  if (call < 0 || (size_t)call >= s.calls.size()) return -1;
  const Call& c = s.calls[call];
  if (out != nullptr && bytes > 0) std::memcpy(out, s.blob.data() + c.blob0, (size_t)bytes < c.blob_bytes ? (size_t)bytes : c.blob_bytes);
  return (int64_t)c.blob_bytes;
}

static int run_segment(ma_block_table_t* t, int backward, int32_t block, uint32_t seed, ma_stream_t stream) {
  if (t == nullptr || block < 0 || block >= kMaxBlocks) return MA_ERR_INVALID_ARG;
  Segment& s = t->seg[backward][block];
  if (s.calls.empty()) return MA_ERR_INVALID_ARG;  // a block that was never recorded: the caller's table is not the one it thinks
  for (size_t k = 0; k < s.calls.size(); ++k) {
    const Call& c = s.calls[k];
    Ctx ctx{s.blob.data() + c.blob0, c.blob_bytes};
    const int rc = issue(c.fn, s.words.data() + c.word0, ctx, seed, stream);
    if (rc != MA_OK) {
      t->failed_call = (int32_t)k;
      return rc;
    }
  }
  return MA_OK;
}

int ma_conformer_block_fwd_train(ma_block_table_t* table, int32_t block, uint32_t seed, ma_stream_t stream) {
  return run_segment(table, 0, block, seed, stream);
}

int ma_conformer_block_bwd_train(ma_block_table_t* table, int32_t block, uint32_t seed, ma_stream_t stream) {
  return run_segment(table, 1, block, seed, stream);
}

}  // extern "C"
