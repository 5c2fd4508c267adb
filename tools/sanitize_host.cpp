// sanitize_host.cpp — AddressSanitizer + UBSan target for the library's HOST code that needs no
// GPU: the YGOPro record emitter (query-buffer parser + option-string grammar, ygo.hip) and the
// history-ring update. GPU sanitizers are not available on this pool, so only host code is
// instrumented (`make sanitize`, see the Makefile: each -fsanitize= sits behind -Xarch_host).
//
// Inputs: synthetic duel states in the reference's data forms (tools/ygo_synth.c), emitted into
// heap blocks of EXACTLY the record stride (so any write past a record is caught), then the same
// states with their option / revealed strings garbled, truncated or replaced by strings of the
// wrong message type, and with their query buffers garbled and truncated: the emitter must return EP_OK or EP_ERR_INVALID and touch nothing it does
// not own. Exit status 0 = no sanitizer report (the sanitizers abort the process otherwise).
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "../include/envpool_amd_ygo.h"

extern "C" {
struct ys_state;
void ys_make_table(int32_t n, uint64_t seed, ygo_card_data* out);
ys_state* ys_gen(uint64_t seed, int64_t index, const ep_ygo_config* cfg, const ygo_card_data* table, int32_t n_table);
void ys_free(ys_state* s);
}

static uint64_t g_s = 0x1234567;
static uint64_t rnd() {
  uint64_t z = (g_s += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

static const int kMsgs[14] = {10, 11, 12, 13, 14, 15, 16, 18, 19, 20, 23, 24, 26, 141};
static const char kAlphabet[] = "trcsmavbeyfhgxdzo0123456789 abcdefgh\x01\x7f\xff";

int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 20000;
  ep_ygo_config cfg;
  memset(&cfg, 0, sizeof(cfg));
  cfg.max_cards = 75; cfg.max_options = 16; cfg.n_history_actions = 16; cfg.max_multi_select = 5;
  cfg.max_episode_steps = 2147483647; cfg.max_batch = 1;
  std::vector<ygo_card_data> table(1000);
  ys_make_table(1000, 0, table.data());
  const int64_t stride = ep_ygo_record_stride(&cfg);
  const int nf = ep_ygo_action_feats(&cfg);
  long ok = 0, invalid = 0, other = 0, pushes = 0;
  for (int i = 0; i < n; ++i) {
    ys_state* s = ys_gen(7, i, &cfg, table.data(), 1000);
    ygo_raw_state* raw = (ygo_raw_state*)s;  // first member of ys_state
    for (int variant = 0; variant < 4; ++variant) {
      if (variant == 1) {  // another message type with the same strings
        raw->msg = kMsgs[rnd() % 14];
      } else if (variant >= 2) {  // garbled / truncated option and revealed strings
        for (int o = 0; o < raw->n_options; ++o) {
          char* str = (char*)raw->options[o];  // 64-byte cells of the generator's own storage
          const size_t len = strlen(str);
          const int kind = (int)(rnd() % 4);
          if (kind == 0 && len) str[rnd() % len] = kAlphabet[rnd() % (sizeof(kAlphabet) - 1)];
          else if (kind == 1) str[rnd() % (len + 1)] = 0;
          else if (kind == 2) { const size_t m = rnd() % 63; for (size_t k = 0; k < m; ++k) str[k] = kAlphabet[rnd() % (sizeof(kAlphabet) - 1)]; str[m] = 0; }
        }
        for (int o = 0; o < raw->n_revealed; ++o) {
          char* str = (char*)raw->revealed[o];
          const size_t len = strlen(str);
          if (len && rnd() % 2) str[rnd() % len] = kAlphabet[rnd() % (sizeof(kAlphabet) - 1)];
        }
        if (variant == 3) raw->ha_p[rnd() % 2] = (int32_t)(rnd() % 16);
      }
      // variant 3 also garbles / truncates the query_field_card buffers, each moved into a heap block
      // of exactly its length so that a read past the end is caught
      std::vector<uint8_t*> owned;
      const uint8_t* keep_buf[2][7];
      int32_t keep_len[2][7];
      memcpy(keep_buf, raw->field_buf, sizeof(keep_buf));
      memcpy(keep_len, raw->field_len, sizeof(keep_len));
      if (variant == 3) {
        for (int pl = 0; pl < 2; ++pl)
          for (int li = 0; li < 7; ++li) {
            int32_t len = raw->field_len[pl][li];
            if (!raw->field_buf[pl][li] || len <= 0) continue;
            if (rnd() % 3 == 0) len = (int32_t)(rnd() % (uint64_t)(len + 1));  // truncated anywhere
            uint8_t* blk = (uint8_t*)malloc((size_t)(len ? len : 1));
            memcpy(blk, raw->field_buf[pl][li], (size_t)len);
            const int hits = (int)(rnd() % 4);
            for (int h = 0; h < hits && len; ++h) {
              const size_t at = (size_t)(rnd() % (uint64_t)len);
              blk[at] = rnd() % 2 ? 0xff : (uint8_t)rnd();  // (0xff in a count word: "4 billion overlay units")
            }
            owned.push_back(blk);
            raw->field_buf[pl][li] = blk;
            raw->field_len[pl][li] = len;
          }
      }
      void* rec = malloc((size_t)stride);
      const int rc = ep_ygo_emit_record(&cfg, raw, rec);
      memcpy(raw->field_buf, keep_buf, sizeof(keep_buf));
      memcpy(raw->field_len, keep_len, sizeof(keep_len));
      for (uint8_t* b : owned) free(b);
      if (rc == EP_OK) {
        ++ok;
        const int64_t used = ep_ygo_record_used_bytes(&cfg, rec);
        if (used <= 0 || used > stride || (used & 15)) { fprintf(stderr, "bad used bytes %lld\n", (long long)used); return 2; }
      } else if (rc == EP_ERR_INVALID) {
        ++invalid;
      } else {
        ++other;
      }
      free(rec);
      // the chosen option goes into the player's ring (update_history_actions)
      if (raw->n_options > 0) {
        uint8_t* ring = (uint8_t*)malloc((size_t)cfg.n_history_actions * nf);
        memcpy(ring, raw->history[raw->to_play & 1], (size_t)cfg.n_history_actions * nf);
        int32_t hp = raw->ha_p[raw->to_play & 1];
        uint16_t ids[5] = {(uint16_t)rnd(), (uint16_t)rnd(), 0, 0, 0};
        (void)ep_ygo_history_push(&cfg, ring, &hp, raw->msg, raw->options[rnd() % (uint64_t)raw->n_options], ids,
                                  (int32_t)(rnd() % 6));
        if (hp < 0 || hp >= cfg.n_history_actions) { fprintf(stderr, "ring pointer out of range %d\n", hp); return 3; }
        free(ring);
        ++pushes;
      }
    }
    ys_free(s);
  }
  printf("sanitize_host: %d states x 4 variants: %ld ok, %ld rejected as invalid, %ld other; %ld ring pushes; no sanitizer report\n",
         n, ok, invalid, other, pushes);
  return other ? 4 : 0;
}
