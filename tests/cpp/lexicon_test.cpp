/*
 * tests/cpp/lexicon_test.cpp -- dictionary/Utils.h of the facade against the
 * dump the REFERENCE produced for the same files (tests/golden/decodertest/
 * lexicon_dump.txt: loadWords -> createWordDict -> Dictionary(letters) + "<1>"
 * -> tkn2Idx, DecoderTest.cpp:94-98,137-146).  Host only.
 *   lexicon_test <words.lst> <letters.lst> <lexicon_dump.txt>
 */
#include <fstream>
#include <iostream>
#include <sstream>

#include "flashlight/lib/text/dictionary/Utils.h"

using namespace fl::lib::text;

int main(int argc, char** argv) {
  if (argc < 4) {
    return 2;
  }
  auto lexicon = loadWords(argv[1]);
  Dictionary tokenDict = loadDictionary(argv[2]);
  tokenDict.addEntry("<1>");
  auto wordDict = createWordDict(lexicon);
  std::ostringstream os;
  os << tokenDict.indexSize() << ' ' << wordDict.indexSize() << ' ' << tokenDict.getIndex("|") << ' '
     << wordDict.getIndex(kUnkToken) << '\n';
  for (const auto& it : lexicon) {
    const int usrIdx = wordDict.getIndex(it.first);
    for (const auto& tokens : it.second) {
      auto idx = tkn2Idx(tokens, tokenDict, 1);
      os << usrIdx << '\t' << it.first << '\t';
      for (size_t i = 0; i < idx.size(); ++i) {
        os << (i ? " " : "") << idx[i];
      }
      os << '\n';
    }
  }
  os << "#words\n";
  for (size_t i = 0; i < wordDict.indexSize(); ++i) {
    os << wordDict.getEntry((int)i) << '\n';
  }
  std::ifstream ref(argv[3]);
  std::stringstream want;
  want << ref.rdbuf();
  Dictionary rep;
  for (const char* t : {"a", "b", "c", "<1>", "<2>"}) {
    rep.addEntry(t);
  }
  auto packed = packReplabels({0, 1, 1, 2, 2, 2}, rep, 2);
  const bool packOk = packed == std::vector<int>({0, 1, 3, 2, 4});
  const bool same = os.str() == want.str();
  std::cout << (same ? "dump identical" : "dump DIFFERS") << ", packReplabels " << (packOk ? "ok" : "WRONG")
            << ", utf8 " << splitWrd("a\xc3\xa9z").size() << "\n";
  return (same && packOk && splitWrd("a\xc3\xa9z").size() == 3) ? 0 : 1;
}
