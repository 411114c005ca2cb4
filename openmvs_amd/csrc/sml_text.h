// sml_text.h -- the two text conventions of the reference's option / list files, shared by mvs_front.cpp and opt_dense.cpp (host code, no GPU).
#pragma once
#include <stdio.h>
#include <string>
#include <utility>
#include <vector>

namespace smltext {
// The words of one line: blanks separate, double quotes group (Util::CommandLineToArgvA, libs/Common/Util.cpp:740-803)
inline void splitWords(const std::string& line, std::vector<std::string>& words) {
	words.clear();
	bool quoted = false, inSpace = true;
	for (const char a : line) {
		if (quoted) { if (a == '"') quoted = false; else words.back().push_back(a); continue; }
		if (a == '"') { quoted = true; if (inSpace) words.emplace_back(); inSpace = false; }
		else if (a == ' ' || a == '\t' || a == '\n' || a == '\r') inSpace = true;
		else { if (inSpace) words.emplace_back(); words.back().push_back(a); inSpace = false; }
	}
}
static const char* const kBlank = "\n\r\t ";
inline std::string trimmed(const std::string& t) {
	const size_t a = t.find_first_not_of(kBlank);
	if (a == std::string::npos) return std::string();
	return t.substr(a, t.find_last_not_of(kBlank) - a + 1);
}
// The root values of an SML text (SML::ParseSection / ParseValues, libs/Common/SML.cpp:103-227), the container of both text files read here.  A section's own
// entries are the text outside its child sections -- "[name]" followed by "{ ... }", nested at will -- up to the '}' that closes it; the root section has no '}', so
// a stray one ends the document.  One entry per line: "name = value" names the entry, a line without '=' (or with nothing in front of it) is an unnamed entry whose value
// is the line (SML_AUTOVALUES).  Only the root's entries are collected (neither reader looks into children), in file order as (name, value), name empty for unnamed ones.
// Returns false for what the reference's reader calls a parse error (a child section without a name): what was read before it stays in `out`.
// Checked against the reference's own reader compiled verbatim (oracle/_ref/libref_text.so, tests/test_ref_text.py): identical on every text without brackets and braces
// (4 000 random ones) and on well-formed documents with nested sections; a text with unbalanced brackets, or a '}' directly followed by '[', is read differently by the
// reference's token stream (it puts the bracket back into a buffer it has already consumed) -- neither of the two files ever contains one.
inline void entriesOf(const std::string& chunk, std::vector<std::pair<std::string, std::string>>* out) {
	if (!out) return;
	size_t pos = 0;
	while (pos <= chunk.size()) {
		size_t e = chunk.find('\n', pos);
		if (e == std::string::npos) e = chunk.size();
		const std::string line = trimmed(chunk.substr(pos, e - pos));
		pos = e + 1;
		if (line.empty()) continue;
		const size_t eq = line.find('=');
		const std::string name = eq == std::string::npos ? std::string() : trimmed(line.substr(0, eq));
		if (name.empty()) out->emplace_back(std::string(), eq == std::string::npos ? line : trimmed(line.substr(eq + 1)));
		else out->emplace_back(name, trimmed(line.substr(eq + 1)));
	}
}
// one section starting at text[pos]; returns false on a parse error; pos ends behind the section's '}' (or at the end of the text)
inline bool parseSection(const std::string& text, size_t& pos, std::vector<std::pair<std::string, std::string>>* out, int depth = 0) {
	for (;;) {
		size_t open = text.find('[', pos);
		const bool last = open == std::string::npos;
		if (last) open = text.size();
		const size_t close = text.find('}', pos);
		if (close != std::string::npos && close < open) { entriesOf(text.substr(pos, close - pos), out); pos = close + 1; return true; }   // this section ends here
		entriesOf(text.substr(pos, open - pos), out);
		if (last) { pos = text.size(); return true; }
		const size_t nameEnd = text.find(']', open + 1);
		if (nameEnd == std::string::npos) { pos = text.size(); return true; }                                // a name that is never closed swallows the rest of the text
		if (nameEnd == open + 1) { pos = text.size(); return false; }                                        // "[]": lenName == 0 (blanks count as a name)
		const size_t body = text.find('{', nameEnd + 1);
		if (body == std::string::npos) { pos = text.size(); return true; }
		pos = body + 1;
		if (depth > 64 || !parseSection(text, pos, nullptr, depth + 1)) return false;      // the child's entries are not ours
	}
}
inline bool readAll(const char* path, std::string& text) {
	FILE* f = fopen(path, "rb");
	if (!f) return false;
	text.clear();
	char buf[4096]; size_t k;
	while ((k = fread(buf, 1, sizeof(buf), f)) > 0) text.append(buf, k);
	fclose(f);
	return true;
}
// 0 = read, 1 = parse error (entries in front of it are in `out`), -1 = the file cannot be opened
inline int rootValues(const char* path, std::vector<std::pair<std::string, std::string>>& out) {
	out.clear();
	std::string text;
	if (!readAll(path, text)) return -1;
	size_t pos = 0;
	return parseSection(text, pos, &out) ? 0 : 1;
}
} // namespace smltext
