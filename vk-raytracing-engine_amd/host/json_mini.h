// json_mini.h -- small recursive-descent JSON reader for config.json and glTF 2.0 documents.
// Replaces the reference's vendored nlohmann/json (common/json.hpp, consumer main.cpp:136-145) and
// tinygltf's JSON layer (hello_vulkan.cpp:329-342).  Header only, no dependencies.
#pragma once
#include <cmath>
#include <cstdlib>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

namespace vkrt_host {

class Json
{
public:
  enum Type { Null, Bool, Number, String, Array, Object };
  Type type = Null;
  bool b = false;
  double num = 0;
  std::string str;
  std::vector<Json> arr;
  std::vector<std::pair<std::string, Json>> obj;  // insertion order kept (attribute order matters for caching keys)

  bool isNull() const { return type == Null; }
  bool isObject() const { return type == Object; }
  bool isArray() const { return type == Array; }
  bool isNumber() const { return type == Number; }
  bool isString() const { return type == String; }
  size_t size() const { return type == Array ? arr.size() : type == Object ? obj.size() : 0; }

  const Json* find(const std::string& k) const
  {
    if(type != Object)
      return nullptr;
    for(const auto& kv : obj)
      if(kv.first == k)
        return &kv.second;
    return nullptr;
  }
  bool has(const std::string& k) const { return find(k) != nullptr; }
  const Json& operator[](const std::string& k) const
  {
    static const Json null_;
    const Json* j = find(k);
    return j ? *j : null_;
  }
  const Json& operator[](size_t i) const
  {
    static const Json null_;
    return (type == Array && i < arr.size()) ? arr[i] : null_;
  }
  double number(double dflt = 0) const { return type == Number ? num : dflt; }
  int integer(int dflt = 0) const { return type == Number ? (int)std::llround(num) : dflt; }
  bool boolean(bool dflt = false) const { return type == Bool ? b : dflt; }
  std::string string(const std::string& dflt = "") const { return type == String ? str : dflt; }

  static Json parse(const std::string& text)
  {
    Parser p{text, 0};
    p.ws();
    Json j = p.value();
    p.ws();
    if(p.i != text.size())
      throw std::runtime_error("json: trailing characters at offset " + std::to_string(p.i));
    return j;
  }

private:
  struct Parser
  {
    const std::string& s;
    size_t i;
    int depth = 0;  // open containers around the value being parsed: recursion is bounded (a file of 100 k '[' overflowed the stack;
                    // found by the sanitizer build, tests/test_host_asan.py).  glTF needs a dozen levels; tinygltf's json.hpp has a limit too
    static constexpr int kMaxDepth = 192;
    struct Nest
    {
      Parser& p;
      explicit Nest(Parser& q) : p(q) { if(++p.depth > kMaxDepth) p.err("nesting too deep"); }
      ~Nest() { p.depth--; }
    };
    [[noreturn]] void err(const char* m) { throw std::runtime_error(std::string("json: ") + m + " at offset " + std::to_string(i)); }
    void ws()
    {
      while(i < s.size() && (s[i] == ' ' || s[i] == '\t' || s[i] == '\n' || s[i] == '\r'))
        i++;
    }
    bool lit(const char* w)
    {
      size_t n = 0;
      while(w[n]) n++;
      if(s.compare(i, n, w) == 0) { i += n; return true; }
      return false;
    }
    Json value()
    {
      if(i >= s.size()) err("unexpected end");
      Json j;
      const char c = s[i];
      if(c == '{')
      {
        const Nest nest(*this);
        j.type = Object;
        i++;
        ws();
        if(i < s.size() && s[i] == '}') { i++; return j; }
        for(;;)
        {
          ws();
          if(i >= s.size() || s[i] != '"') err("expected string key");
          std::string k = str();
          ws();
          if(i >= s.size() || s[i] != ':') err("expected ':'");
          i++;
          ws();
          j.obj.emplace_back(std::move(k), value());
          ws();
          if(i < s.size() && s[i] == ',') { i++; continue; }
          if(i < s.size() && s[i] == '}') { i++; break; }
          err("expected ',' or '}'");
        }
      }
      else if(c == '[')
      {
        const Nest nest(*this);
        j.type = Array;
        i++;
        ws();
        if(i < s.size() && s[i] == ']') { i++; return j; }
        for(;;)
        {
          ws();
          j.arr.push_back(value());
          ws();
          if(i < s.size() && s[i] == ',') { i++; continue; }
          if(i < s.size() && s[i] == ']') { i++; break; }
          err("expected ',' or ']'");
        }
      }
      else if(c == '"')
      {
        j.type = String;
        j.str = str();
      }
      else if(lit("true")) { j.type = Bool; j.b = true; }
      else if(lit("false")) { j.type = Bool; j.b = false; }
      else if(lit("null")) { j.type = Null; }
      else
      {
        const char* st = s.c_str() + i;
        char* en = nullptr;
        j.num = std::strtod(st, &en);
        if(en == st) err("bad value");
        j.type = Number;
        i += (size_t)(en - st);
      }
      return j;
    }
    static void utf8(std::string& o, unsigned cp)
    {
      if(cp < 0x80) o += (char)cp;
      else if(cp < 0x800) { o += (char)(0xC0 | (cp >> 6)); o += (char)(0x80 | (cp & 0x3F)); }
      else if(cp < 0x10000) { o += (char)(0xE0 | (cp >> 12)); o += (char)(0x80 | ((cp >> 6) & 0x3F)); o += (char)(0x80 | (cp & 0x3F)); }
      else { o += (char)(0xF0 | (cp >> 18)); o += (char)(0x80 | ((cp >> 12) & 0x3F)); o += (char)(0x80 | ((cp >> 6) & 0x3F)); o += (char)(0x80 | (cp & 0x3F)); }
    }
    std::string str()
    {
      std::string o;
      i++;  // opening quote
      while(i < s.size() && s[i] != '"')
      {
        char c = s[i++];
        if(c != '\\') { o += c; continue; }
        if(i >= s.size()) err("bad escape");
        c = s[i++];
        switch(c)
        {
          case '"': o += '"'; break;
          case '\\': o += '\\'; break;
          case '/': o += '/'; break;
          case 'b': o += '\b'; break;
          case 'f': o += '\f'; break;
          case 'n': o += '\n'; break;
          case 'r': o += '\r'; break;
          case 't': o += '\t'; break;
          case 'u':
          {
            if(i + 4 > s.size()) err("bad \\u escape");
            unsigned cp = (unsigned)std::strtoul(s.substr(i, 4).c_str(), nullptr, 16);
            i += 4;
            if(cp >= 0xD800 && cp < 0xDC00 && i + 6 <= s.size() && s[i] == '\\' && s[i + 1] == 'u')
            {
              unsigned lo = (unsigned)std::strtoul(s.substr(i + 2, 4).c_str(), nullptr, 16);
              cp = 0x10000 + ((cp - 0xD800) << 10) + (lo - 0xDC00);
              i += 6;
            }
            utf8(o, cp);
            break;
          }
          default: err("bad escape");
        }
      }
      if(i >= s.size()) err("unterminated string");
      i++;  // closing quote
      return o;
    }
  };
};

}  // namespace vkrt_host
