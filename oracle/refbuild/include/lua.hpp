// oracle/refbuild/include/lua.hpp -- TEST INFRASTRUCTURE ONLY.
// A host-side stand-in for the handful of Lua C-API calls that the reference's core/lua_calls.h makes, so that the
// real reference sources compile without Lua/Torch (un-vendored third-party dependencies).  A lua_State here is a
// value stack plus a table of named HOST callbacks: lua_getglobal pushes a callback, lua_pcall runs it on the
// arguments pushed after it.  Tables are 1-based arrays of numbers, which is all lua_calls.h ever builds or reads.
// oracle/refbuild/ref_glue.cpp installs "forward"/"backward" callbacks that play the role of the Torch scripts
// (scene-coordinate CNN = a stored prediction, score CNN = the soft-inlier score and its analytic gradient).
#pragma once
#include <cstdio>
#include <functional>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#define LUA_MULTRET (-1)

struct LuaValue {
    enum Kind { NIL, NUM, STR, TABLE, FUNC } kind = NIL;
    double num = 0;
    std::string str;                            // STR: text, FUNC: global name
    std::shared_ptr<std::vector<double>> tab;  // TABLE: element i (1-based) at (*tab)[i-1]
    static LuaValue number(double v) { LuaValue x; x.kind = NUM; x.num = v; return x; }
    static LuaValue string(const std::string& s) { LuaValue x; x.kind = STR; x.str = s; return x; }
    static LuaValue table(size_t reserve = 0) { LuaValue x; x.kind = TABLE; x.tab = std::make_shared<std::vector<double>>(); x.tab->reserve(reserve); return x; }
};

struct lua_State {
    typedef std::function<std::vector<LuaValue>(std::vector<LuaValue>&)> Callback;
    std::vector<LuaValue> stack;
    std::map<std::string, Callback> globals;
    int calls = 0;
};

static inline LuaValue& lua_at(lua_State* L, int idx) {
    const int n = (int)L->stack.size();
    const int i = idx > 0 ? idx - 1 : n + idx;
    if (i < 0 || i >= n) throw std::runtime_error("mini-lua: bad stack index");
    return L->stack[i];
}
static inline int lua_gettop(lua_State* L) { return (int)L->stack.size(); }
static inline void lua_pop(lua_State* L, int n) { L->stack.resize(L->stack.size() - n); }
static inline void lua_pushnumber(lua_State* L, double v) { L->stack.push_back(LuaValue::number(v)); }
static inline void lua_pushinteger(lua_State* L, long v) { L->stack.push_back(LuaValue::number((double)v)); }
static inline void lua_pushstring(lua_State* L, const char* s) { L->stack.push_back(LuaValue::string(s)); }
static inline void lua_createtable(lua_State* L, int narr, int) { L->stack.push_back(LuaValue::table(narr > 0 ? (size_t)narr : 0)); }
static inline void lua_rawseti(lua_State* L, int idx, int n) {  // t[n] = top; pop
    LuaValue& t = lua_at(L, idx);
    if (t.kind != LuaValue::TABLE || n < 1) throw std::runtime_error("mini-lua: rawseti on a non-table");
    if ((size_t)n > t.tab->size()) t.tab->resize(n, 0.0);
    (*t.tab)[n - 1] = L->stack.back().num;
    L->stack.pop_back();
}
static inline void lua_gettable(lua_State* L, int idx) {  // key = top; replace it by t[key]
    LuaValue& t = lua_at(L, idx);
    if (t.kind != LuaValue::TABLE) throw std::runtime_error("mini-lua: gettable on a non-table");
    const long k = (long)L->stack.back().num;
    const double v = (k >= 1 && (size_t)k <= t.tab->size()) ? (*t.tab)[k - 1] : 0.0;
    L->stack.back() = LuaValue::number(v);
}
static inline double lua_tonumber(lua_State* L, int idx) { return lua_at(L, idx).num; }
static inline const char* lua_tostring(lua_State* L, int idx) { return lua_at(L, idx).str.c_str(); }
static inline void lua_getglobal(lua_State* L, const char* name) { LuaValue f; f.kind = LuaValue::FUNC; f.str = name; L->stack.push_back(f); }
static inline int luaL_loadfile(lua_State* L, const char* filename) { L->stack.push_back(LuaValue::string(std::string("mini-lua cannot load ") + filename)); return 1; }
static inline int lua_pcall(lua_State* L, int nargs, int nresults, int) {
    const int n = (int)L->stack.size();
    if (n < nargs + 1) throw std::runtime_error("mini-lua: pcall stack underflow");
    LuaValue fn = L->stack[n - nargs - 1];
    std::vector<LuaValue> args(L->stack.begin() + (n - nargs), L->stack.end());
    L->stack.resize(n - nargs - 1);
    auto it = L->globals.find(fn.str);
    if (fn.kind != LuaValue::FUNC || it == L->globals.end()) throw std::runtime_error("mini-lua: call of an undefined global '" + fn.str + "'");
    L->calls++;
    std::vector<LuaValue> res = it->second(args);
    if (nresults != LUA_MULTRET) res.resize(nresults);
    for (auto& r : res) L->stack.push_back(r);
    return 0;
}
