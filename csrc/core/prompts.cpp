#include "prompts.hpp"

#include <algorithm>

namespace lc {

std::vector<Persona> default_personas() {
  return {
      {"High Society", "Society and Culture",
       "\n* Shared norms, values and belief systems"
       "\n* The historical background of events"
       "\n* Cultural variety and tradition"
       "\n* Institutions that structure society (family, schooling, government)"
       "\n* Effects on how people behave and interact"
       "\n* Ethics and moral weight"
       "\n* Present-day events and social debates"
       "\n* Population trends and demographics"
       "\n* Language and styles of communication"
       "\n* Art, literature and folklore as mirrors of a society"},
      {"The Technician", "Technical Detail",
       "\n* Whether the information is accurate and precise"
       "\n* Concrete measurements, quantities and units"
       "\n* Technical standards and specifications"
       "\n* Procedures and processes described in detail"
       "\n* Underlying scientific principles and theory"
       "\n* Formulas, equations and the mathematics involved"
       "\n* Sound logic and problem-solving"
       "\n* Cause and effect"
       "\n* Explanations and instructions given step by step"
       "\n* Completeness and attention to detail"},
      {"Art Boy", "Art and Imagination",
       "\n* Creative expression in any medium (visual, sound, writing and others)"
       "\n* Tools and techniques for making art, digital or traditional"
       "\n* Using art to explore emotion, ideas and concepts"
       "\n* Imagination, originality and innovation"
       "\n* Aesthetic principles such as composition, colour and form"
       "\n* Art history, movements and styles"
       "\n* How culture and society shape art"
       "\n* Visualising data or building simulations for artistic ends"
       "\n* Interactive works and installations"
       "\n* Art as a means of communication and storytelling"},
      {"Programming Nerd", "Computer Science",
       "\n* Data structures and algorithms"
       "\n* Programming languages and their paradigms"
       "\n* Principles of software engineering"
       "\n* Hardware and computer architecture"
       "\n* Networks and distributed systems"
       "\n* Machine learning and artificial intelligence"
       "\n* Security and the privacy of data"
       "\n* Complexity and the theory of computation"
       "\n* Databases and managing data"
       "\n* Operating systems and systems programming"},
  };
}

PromptTemplates PromptTemplates::defaults() {
  PromptTemplates t;
  t.propose =
      "Answer the question below. Do not describe yourself as a language model.\n\n{question}";
  t.evaluate =
      "\n---\nQuestion: {question}\n---\nAnswer: {answer}\n---\nYour Instructions:\n"
      "You belong to a panel of LLMs that must agree on an answer to the question above. "
      "The answer shown came from the first panel member chosen. Judge that answer from the "
      "point of view of your knowledge domain: {domain}. You may reply with only one of two "
      "verdicts: Good or NeedsRefinement.\n\n"
      "A direct link between the answer and your domain is not required; think about indirect "
      "or tangential links too, and about how the answer could enable, inspire or be used in "
      "work belonging to your domain. In particular keep these aspects in mind:{tuning}\n\n"
      "First decide whether the question has anything to do with your domain. If it does not, "
      "you are not qualified to judge, so reply with exactly Good. If it does and you find the "
      "answer good, reply with exactly Good. If you find the answer bad, reply with exactly "
      "NeedsRefinement. Then, on a new line, give the reasoning behind your verdict.\n"
      "---\nExamples:\n\n"
      "Question: Which programming language suits a beginner?\n"
      "Answer: Python\n"
      "Your domain: art and imagination\n"
      "Evaluation: Good\n"
      "Reasoning: The question is outside your domain.\n\n"
      "Question: What makes software simpler to update?\n"
      "Answer: Decoupling\n"
      "Your domain: technical rigor\n"
      "Evaluation: NeedsRefinement\n"
      "Reasoning: Loose coupling is only one ingredient of maintainable software and the "
      "answer gives too little detail.";
  t.refine =
      "\n---\nQuestion: {question}\n---\nAnswer: {answer}\n---\nYour Instructions:\n"
      "A user asked the question above and was given the answer shown. When you evaluated that "
      "answer you said it needed refinement. Refine it as your knowledge domain, {domain}, "
      "requires.{reasoning}\n\n"
      "Let the following points guide the refinement; they need not appear in the answer "
      "itself:{tuning}";
  return t;
}

PromptTemplates PromptTemplates::persona_last() {
  PromptTemplates t = defaults();
  t.evaluate =
      "\n---\nQuestion: {question}\n---\nAnswer: {answer}\n---\nYour Instructions:\n"
      "You belong to a panel of LLMs that must agree on an answer to the question above. "
      "The answer shown came from the first panel member chosen. Judge that answer from the "
      "point of view of your knowledge domain, which is named at the end of these instructions. "
      "You may reply with only one of two verdicts: Good or NeedsRefinement.\n\n"
      "A direct link between the answer and your domain is not required; think about indirect "
      "or tangential links too, and about how the answer could enable, inspire or be used in "
      "work belonging to your domain.\n\n"
      "First decide whether the question has anything to do with your domain. If it does not, "
      "you are not qualified to judge, so reply with exactly Good. If it does and you find the "
      "answer good, reply with exactly Good. If you find the answer bad, reply with exactly "
      "NeedsRefinement. Then, on a new line, give the reasoning behind your verdict.\n"
      "---\nExamples:\n\n"
      "Question: Which programming language suits a beginner?\n"
      "Answer: Python\n"
      "Your domain: art and imagination\n"
      "Evaluation: Good\n"
      "Reasoning: The question is outside your domain.\n\n"
      "Question: What makes software simpler to update?\n"
      "Answer: Decoupling\n"
      "Your domain: technical rigor\n"
      "Evaluation: NeedsRefinement\n"
      "Reasoning: Loose coupling is only one ingredient of maintainable software and the "
      "answer gives too little detail.\n"
      "---\nYour knowledge domain: {domain}. In particular keep these aspects in mind:{tuning}";
  return t;
}

PromptTemplates PromptTemplates::persona_first() {
  PromptTemplates t = defaults();
  const std::string head =
      "\n---\nYour Instructions:\n"
      "You belong to a panel of LLMs that must agree on an answer to a question. The question and "
      "the answer, which came from the first panel member chosen, are shown at the end of these "
      "instructions. ";
  const std::string rules =
      "A direct link between the answer and your domain is not required; think about indirect "
      "or tangential links too, and about how the answer could enable, inspire or be used in "
      "work belonging to your domain. In particular keep these aspects in mind:{tuning}\n\n";
  const std::string tail = "\n---\nQuestion: {question}\n---\nAnswer: {answer}";
  t.evaluate = head +
      "Judge the answer from the point of view of your knowledge domain: {domain}. You may reply "
      "with only one of two verdicts: Good or NeedsRefinement.\n\n" + rules +
      "First decide whether the question has anything to do with your domain. If it does not, "
      "you are not qualified to judge, so reply with exactly Good. If it does and you find the "
      "answer good, reply with exactly Good. If you find the answer bad, reply with exactly "
      "NeedsRefinement. Then, on a new line, give the reasoning behind your verdict.\n"
      "---\nExamples:\n\n"
      "Question: Which programming language suits a beginner?\n"
      "Answer: Python\n"
      "Your domain: art and imagination\n"
      "Evaluation: Good\n"
      "Reasoning: The question is outside your domain.\n\n"
      "Question: What makes software simpler to update?\n"
      "Answer: Decoupling\n"
      "Your domain: technical rigor\n"
      "Evaluation: NeedsRefinement\n"
      "Reasoning: Loose coupling is only one ingredient of maintainable software and the "
      "answer gives too little detail." + tail;
  t.refine =
      "\n---\nYour Instructions:\n"
      "A user asked a question and was given an answer; both are shown at the end of these "
      "instructions. When you evaluated that answer you said it needed refinement. Refine it as "
      "your knowledge domain, {domain}, requires.\n\n"
      "Let the following points guide the refinement; they need not appear in the answer "
      "itself:{tuning}" + tail + "{reasoning}";
  return t;
}

namespace {

// Single-pass substitution so that text inserted for one placeholder is never
// re-scanned for another ("{answer}" inside a question stays literal).
std::string substitute(const std::string& tmpl,
                       const std::vector<std::pair<std::string, std::string>>& kv) {
  std::string out;
  out.reserve(tmpl.size() + 256);
  size_t i = 0;
  while (i < tmpl.size()) {
    if (tmpl[i] == '{') {
      bool hit = false;
      for (const auto& p : kv) {
        if (tmpl.compare(i, p.first.size(), p.first) == 0) {
          out += p.second;
          i += p.first.size();
          hit = true;
          break;
        }
      }
      if (hit) continue;
    }
    out.push_back(tmpl[i++]);
  }
  return out;
}

std::string strip_quotes(std::string s) {
  s.erase(std::remove(s.begin(), s.end(), '"'), s.end());
  return s;
}

}  // namespace

std::string render_propose(const PromptTemplates& t, const std::string& question) {
  // The reference does not strip quotes from the propose prompt (main.rs:95).
  return substitute(t.propose, {{"{question}", question}});
}

std::string render_evaluate(const PromptTemplates& t, const std::string& question,
                            const std::string& answer, const Persona& p) {
  std::string s = substitute(t.evaluate, {{"{question}", question},
                                          {"{answer}", answer},
                                          {"{domain}", p.domain},
                                          {"{tuning}", p.tuning}});
  return t.strip_quotes ? strip_quotes(std::move(s)) : s;
}

std::string render_refine(const PromptTemplates& t, const std::string& question,
                          const std::string& answer, const Persona& p,
                          const std::string& reasoning) {
  std::string extra;
  if (!reasoning.empty()) extra = "\n\nThe panel's objections were:\n" + reasoning;
  std::string s = substitute(t.refine, {{"{question}", question},
                                        {"{answer}", answer},
                                        {"{domain}", p.domain},
                                        {"{tuning}", p.tuning},
                                        {"{reasoning}", extra}});
  return t.strip_quotes ? strip_quotes(std::move(s)) : s;
}

std::string evaluate_shared_prefix(const PromptTemplates& t, const std::string& question,
                                   const std::string& answer) {
  size_t cut = std::min(t.evaluate.find("{domain}"), t.evaluate.find("{tuning}"));
  std::string head = cut == std::string::npos ? t.evaluate : t.evaluate.substr(0, cut);
  std::string s = substitute(head, {{"{question}", question}, {"{answer}", answer}});
  return t.strip_quotes ? strip_quotes(std::move(s)) : s;
}

std::string render_multiple_choice(const std::string& question,
                                   const std::vector<std::string>& options,
                                   const std::string& choice_letters, bool chain_of_thought) {
  std::string s = "Question: " + question + "\n";
  for (size_t i = 0; i < options.size() && i < choice_letters.size(); ++i) {
    s.push_back(choice_letters[i]);
    s += ". " + options[i] + "\n";
  }
  if (chain_of_thought)
    s += "Think step by step, then finish with a line of the form \"Answer: X\".\n";
  else
    s += "Answer:";
  return s;
}

}  // namespace lc
