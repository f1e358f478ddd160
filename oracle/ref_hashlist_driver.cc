// oracle/ref_hashlist_driver.cc -- thin C wrapper around the REFERENCE's own
// HashList<int32_t,int32_t>, compiled in place from /root/reference (never
// copied) into oracle/_ref/libref_hashlist.so by oracle/Makefile.
//
// TEST INFRASTRUCTURE ONLY: tests/test_oracle_hashlist.py uses it to pin the
// iteration order of the oracle's HashList restatement (oracle/kd_oracle.cc),
// which is what makes the reference's pruning order dependent
// (/root/reference/kaldi-decoder/csrc/hash-list-inl.h:128-173).
// hash-list.h is header-only and needs only stl-utils.h and log.h, both in the
// reference tree, so this is the one part of the hot path that builds here.

#include <cstdint>

#include "kaldi-decoder/csrc/hash-list.h"

using HL = kaldi_decoder::HashList<int32_t, int32_t>;

extern "C" {

void *ref_hashlist_create() { return new HL; }
void ref_hashlist_free(void *h) { delete static_cast<HL *>(h); }
void ref_hashlist_set_size(void *h, int64_t n) { static_cast<HL *>(h)->SetSize(static_cast<size_t>(n)); }

int32_t ref_hashlist_insert(void *hv, int32_t key, int32_t val) {
  HL::Elem *e = static_cast<HL *>(hv)->Insert(key, val);
  return e->val == val ? 1 : 0;
}

int32_t ref_hashlist_find(void *hv, int32_t key) {
  HL::Elem *e = static_cast<HL *>(hv)->Find(key);
  return e == nullptr ? -1 : e->val;
}

int32_t ref_hashlist_clear(void *hv, int32_t cap, int32_t *keys, int32_t *vals) {
  HL *h = static_cast<HL *>(hv);
  int32_t n = 0;
  for (HL::Elem *e = h->Clear(), *nx; e != nullptr; e = nx) {
    nx = e->tail;
    if (n < cap) { keys[n] = e->key; vals[n] = e->val; }
    n++;
    h->Delete(e);
  }
  return n;
}

int32_t ref_hashlist_list(void *hv, int32_t cap, int32_t *keys, int32_t *vals) {
  int32_t n = 0;
  for (const HL::Elem *e = static_cast<HL *>(hv)->GetList(); e != nullptr; e = e->tail) {
    if (n < cap) { keys[n] = e->key; vals[n] = e->val; }
    n++;
  }
  return n;
}

}  // extern "C"
