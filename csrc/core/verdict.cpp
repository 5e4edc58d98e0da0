#include "verdict.hpp"

#include <algorithm>
#include <cctype>
#include <cstdlib>

namespace lc {

const char* verdict_name(Verdict v) {
  switch (v) {
    case Verdict::Good: return "Good";
    case Verdict::NeedsRefinement: return "NeedsRefinement";
    default: return "Abstain";
  }
}

namespace {

std::vector<std::string> nonempty_lines(const std::string& s) {
  std::vector<std::string> out;
  size_t start = 0;
  while (start <= s.size()) {
    size_t nl = s.find('\n', start);
    if (nl == std::string::npos) nl = s.size();
    if (nl > start) out.emplace_back(s.substr(start, nl - start));
    start = nl + 1;
  }
  return out;
}

std::string join_paragraphs(const std::vector<std::string>& parts, size_t from) {
  std::string out;
  for (size_t i = from; i < parts.size(); ++i) {
    if (i > from) out += "\n\n";
    out += parts[i];
  }
  return out;
}

std::string lower(std::string s) {
  for (auto& c : s) c = static_cast<char>(std::tolower(static_cast<unsigned char>(c)));
  return s;
}

// Keeps letters only, lower-cased: "**Needs Refinement.**" -> "needsrefinement".
std::string letters_only(const std::string& s) {
  std::string out;
  for (unsigned char c : s)
    if (std::isalpha(c)) out.push_back(static_cast<char>(std::tolower(c)));
  return out;
}

bool starts_with(const std::string& s, const char* p) { return s.rfind(p, 0) == 0; }

}  // namespace

ParsedVerdict parse_verdict(const std::string& reply, ParseMode mode) {
  ParsedVerdict r;
  std::vector<std::string> parts = nonempty_lines(reply);
  if (parts.empty()) return r;  // reference would panic here; we degrade to NeedsRefinement

  // Strict rule: remove spaces (only U+0020, like the reference's replace(" ", "")).
  std::string head = parts[0];
  head.erase(std::remove(head.begin(), head.end(), ' '), head.end());
  r.reasoning = join_paragraphs(parts, 1);
  if (head == "Good") {
    r.verdict = Verdict::Good;
    r.well_formed = true;
    return r;
  }
  if (head == "NeedsRefinement") {
    r.verdict = Verdict::NeedsRefinement;
    r.well_formed = true;
    return r;
  }
  if (mode == ParseMode::Strict) return r;

  // Tolerant: strip an "Evaluation:"/"Verdict:"/"Answer:" label, then look at the
  // leading word(s) of what remains; the rest of that line joins the reasoning.
  std::string line = parts[0];
  std::string ll = lower(line);
  for (const char* label : {"evaluation", "verdict", "answer", "feedback", "decision"}) {
    size_t p = ll.find(label);
    if (p != std::string::npos && p < 8) {
      size_t colon = ll.find_first_of(":-=", p);
      if (colon != std::string::npos) {
        line = line.substr(colon + 1);
        ll = lower(line);
      }
      break;
    }
  }
  std::string key = letters_only(line);
  std::string same_line_reasoning;
  auto take_rest = [&](size_t n_letters) {
    // reasoning that followed the verdict word on the same line
    size_t seen = 0, i = 0;
    for (; i < line.size() && seen < n_letters; ++i)
      if (std::isalpha(static_cast<unsigned char>(line[i]))) ++seen;
    while (i < line.size() && !std::isalnum(static_cast<unsigned char>(line[i]))) ++i;
    if (i < line.size()) same_line_reasoning = line.substr(i);
  };
  bool matched = false;
  if (starts_with(key, "needsrefinement")) {
    r.verdict = Verdict::NeedsRefinement;
    take_rest(15);
    matched = true;
  } else if (starts_with(key, "good")) {
    r.verdict = Verdict::Good;
    take_rest(4);
    matched = true;
  }
  if (matched) {
    r.well_formed = true;
    // "Reasoning: ..." label on the following line is kept verbatim (it is content).
    if (!same_line_reasoning.empty()) {
      r.reasoning = r.reasoning.empty() ? same_line_reasoning
                                        : same_line_reasoning + "\n\n" + r.reasoning;
    }
  }
  return r;
}

int extract_choice(const std::string& text, const std::string& choices) {
  if (choices.empty() || text.empty()) return -1;
  auto idx_of = [&](char c) -> int {
    char u = static_cast<char>(std::toupper(static_cast<unsigned char>(c)));
    for (size_t i = 0; i < choices.size(); ++i)
      if (std::toupper(static_cast<unsigned char>(choices[i])) == u) return static_cast<int>(i);
    return -1;
  };
  auto is_word_char = [](char c) { return std::isalnum(static_cast<unsigned char>(c)) || c == '_'; };
  std::string ll = lower(text);
  int best = -1;
  // 1) "answer is X", "answer: X", "answer - (X)" ; keep the LAST occurrence.
  size_t pos = 0;
  while ((pos = ll.find("answer", pos)) != std::string::npos) {
    size_t i = pos + 6;
    while (i < ll.size() && (ll[i] == ' ' || ll[i] == ':' || ll[i] == '-' || ll[i] == '*' ||
                             ll[i] == '(' || ll[i] == '[' || ll[i] == '\t'))
      ++i;
    if (ll.compare(i, 3, "is ") == 0) {
      i += 3;
      while (i < ll.size() && (ll[i] == ' ' || ll[i] == ':' || ll[i] == '(' || ll[i] == '*' || ll[i] == '['))
        ++i;
    }
    if (i < text.size()) {
      int c = idx_of(text[i]);
      bool boundary = (i + 1 >= text.size()) || !is_word_char(text[i + 1]);
      if (c >= 0 && boundary) best = c;
    }
    pos += 6;
  }
  if (best >= 0) return best;
  // 2) last "(X)" occurrence.
  for (size_t i = 0; i + 2 < text.size(); ++i)
    if (text[i] == '(' && text[i + 2] == ')') {
      int c = idx_of(text[i + 1]);
      if (c >= 0) best = c;
    }
  if (best >= 0) return best;
  // 3) a reply that starts with a bare choice letter ("B", "B.", "B) because").
  size_t i = 0;
  while (i < text.size() && std::isspace(static_cast<unsigned char>(text[i]))) ++i;
  if (i < text.size()) {
    int c = idx_of(text[i]);
    bool boundary = (i + 1 >= text.size()) || !is_word_char(text[i + 1]);
    if (c >= 0 && boundary) return c;
  }
  return -1;
}

std::string normalize_answer(const std::string& text) {
  std::string out;
  bool pending_space = false;
  for (unsigned char c : text) {
    if (std::isspace(c)) {
      pending_space = !out.empty();
      continue;
    }
    if (pending_space) out.push_back(' ');
    pending_space = false;
    out.push_back(static_cast<char>(std::tolower(c)));
  }
  auto is_edge_punct = [](char c) {
    return std::ispunct(static_cast<unsigned char>(c)) != 0;
  };
  size_t b = 0, e = out.size();
  while (b < e && is_edge_punct(out[b])) ++b;
  while (e > b && is_edge_punct(out[e - 1])) --e;
  return out.substr(b, e - b);
}

bool extract_number(const std::string& text, double* out) {
  auto parse_at = [&](size_t i, double* v, size_t* end) -> bool {
    std::string buf;
    size_t j = i;
    if (j < text.size() && (text[j] == '-' || text[j] == '+')) buf.push_back(text[j++]);
    bool digits = false, dot = false;
    for (; j < text.size(); ++j) {
      char c = text[j];
      if (std::isdigit(static_cast<unsigned char>(c))) {
        buf.push_back(c);
        digits = true;
      } else if (c == ',' && digits && j + 1 < text.size() &&
                 std::isdigit(static_cast<unsigned char>(text[j + 1]))) {
        continue;  // thousands separator
      } else if (c == '.' && !dot && j + 1 < text.size() &&
                 std::isdigit(static_cast<unsigned char>(text[j + 1]))) {
        buf.push_back(c);
        dot = true;
      } else {
        break;
      }
    }
    if (!digits) return false;
    *v = std::strtod(buf.c_str(), nullptr);
    *end = j;
    return true;
  };
  size_t marker = text.rfind("####");
  if (marker != std::string::npos) {
    size_t i = marker + 4;
    while (i < text.size() && (text[i] == ' ' || text[i] == '$')) ++i;
    size_t end;
    if (parse_at(i, out, &end)) return true;
  }
  bool found = false;
  size_t i = 0;
  while (i < text.size()) {
    unsigned char c = static_cast<unsigned char>(text[i]);
    bool sign = (c == '-' || c == '+') && i + 1 < text.size() &&
                std::isdigit(static_cast<unsigned char>(text[i + 1])) &&
                (i == 0 || !std::isalnum(static_cast<unsigned char>(text[i - 1])));
    if (std::isdigit(c) || sign) {
      double v;
      size_t end;
      if (parse_at(i, &v, &end)) {
        *out = v;
        found = true;
        i = end;
        continue;
      }
    }
    ++i;
  }
  return found;
}

}  // namespace lc
