// Byte-level fallback tokenizer: every UTF-8 byte is a token, plus BOS/EOS/PAD.
// Lets the whole stack run on random-init checkpoints with no tokenizer files; real
// checkpoints use their Hugging Face tokenizer through the Python adapter
// (llm_consensus_amd/utils/tokenizer.py). The reference needs no tokenizer because
// text goes to a remote API as a string (/root/reference/src/main.rs:84).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace lc {

class ByteTokenizer {
 public:
  static constexpr int32_t kBos = 256;
  static constexpr int32_t kEos = 257;
  static constexpr int32_t kPad = 258;
  static constexpr int32_t kVocab = 259;

  std::vector<int32_t> encode(const std::string& text, bool add_bos) const {
    std::vector<int32_t> ids;
    ids.reserve(text.size() + 1);
    if (add_bos) ids.push_back(kBos);
    for (unsigned char c : text) ids.push_back(static_cast<int32_t>(c));
    return ids;
  }

  // Ids outside the byte range are dropped (specials) or mapped into it (ids a
  // larger-vocabulary random model may emit are folded with modulo 256 when
  // `fold_out_of_range` is set, so decoded text stays printable-ish for demos).
  std::string decode(const std::vector<int32_t>& ids, bool fold_out_of_range) const {
    std::string out;
    out.reserve(ids.size());
    for (int32_t t : ids) {
      if (t >= 0 && t < 256) out.push_back(static_cast<char>(t));
      else if (t >= kVocab && fold_out_of_range) out.push_back(static_cast<char>(t % 256));
    }
    return out;
  }
};

}  // namespace lc
