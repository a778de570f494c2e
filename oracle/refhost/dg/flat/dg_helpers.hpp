// dg_helpers.hpp -- TEST INFRASTRUCTURE ONLY (see dg_mock.hpp).  The DiligentCore *tools* layer the post-process classes use on top of the device: device wrapper
// with / without a state cache, ResourceRegistry, ShaderMacroHelper, the X-suffixed convenience descriptors, scoped debug groups, map helper, the commonly used
// state constants and a no-op Dear ImGui surface (the UpdateUI members of the classes compile, nobody calls them).
#pragma once
#include "dg_mock.hpp"

namespace Diligent
{
// ---- RenderStateCache.hpp
template <bool ThrowOnError = true> class RenderDeviceWithCache
{
public:
    RenderDeviceWithCache(IRenderDevice* pDevice, IRenderStateCache* pCache = nullptr) : m_pDevice{pDevice}, m_pCache{pCache} {}
    RefCntAutoPtr<ITexture> CreateTexture(const TextureDesc& d, const TextureData* data = nullptr)
    {
        RefCntAutoPtr<ITexture> t;
        m_pDevice->CreateTexture(d, data, &t);
        return t;
    }
    RefCntAutoPtr<IBuffer> CreateBuffer(const BufferDesc& d, const BufferData* data = nullptr)
    {
        RefCntAutoPtr<IBuffer> b;
        m_pDevice->CreateBuffer(d, data, &b);
        return b;
    }
    RefCntAutoPtr<IBuffer> CreateBuffer(const BufferDesc& d, const BufferData& data) { return CreateBuffer(d, &data); }
    RefCntAutoPtr<IShader> CreateShader(const ShaderCreateInfo& ci)
    {
        RefCntAutoPtr<IShader> s;
        m_pDevice->CreateShader(ci, &s);
        return s;
    }
    RefCntAutoPtr<IPipelineState> CreateGraphicsPipelineState(const GraphicsPipelineStateCreateInfo& ci)
    {
        RefCntAutoPtr<IPipelineState> p;
        m_pDevice->CreateGraphicsPipelineState(ci, &p);
        return p;
    }
    RefCntAutoPtr<ISampler> CreateSampler(const SamplerDesc& d)
    {
        RefCntAutoPtr<ISampler> s;
        m_pDevice->CreateSampler(d, &s);
        return s;
    }
    IRenderDevice*     GetDevice() const { return m_pDevice; }
    IRenderStateCache* GetCache() const { return m_pCache; }
    operator IRenderDevice*() const { return m_pDevice; }
    IRenderDevice* operator->() const { return m_pDevice; }

private:
    IRenderDevice*     m_pDevice;
    IRenderStateCache* m_pCache;
};
using RenderDeviceWithCache_N = RenderDeviceWithCache<false>;
using RenderDeviceWithCache_E = RenderDeviceWithCache<true>;

// ---- ResourceRegistry.hpp
class ResourceRegistry
{
public:
    using ResourceIdType = Uint32;
    explicit ResourceRegistry(size_t n = 0) : m_Resources(n) {}
    void SetSize(size_t n) { m_Resources.resize(n); }
    void Insert(ResourceIdType id, IDeviceObject* obj)
    {
        if (id >= m_Resources.size()) { Recorder::Get().Error("ResourceRegistry::Insert: id out of range"); return; }
        m_Resources[id] = obj;
    }
    class ResourceAccessor
    {
    public:
        explicit ResourceAccessor(RefCntAutoPtr<IDeviceObject>& r) : m_r{r} {}
        ITexture* AsTexture() const { return dynamic_cast<ITexture*>(m_r.RawPtr()); }
        IBuffer*  AsBuffer() const { return dynamic_cast<IBuffer*>(m_r.RawPtr()); }
        ITextureView* GetTextureSRV() const { return View(TEXTURE_VIEW_SHADER_RESOURCE); }
        ITextureView* GetTextureRTV() const { return View(TEXTURE_VIEW_RENDER_TARGET); }
        ITextureView* GetTextureDSV() const { return View(TEXTURE_VIEW_DEPTH_STENCIL); }
        ITextureView* GetTextureUAV() const { return View(TEXTURE_VIEW_UNORDERED_ACCESS); }
        void          Release() { m_r.Release(); }
        explicit      operator bool() const { return m_r != nullptr; }
        bool          operator!() const { return m_r == nullptr; }
        operator IDeviceObject*() const { return m_r; }
        operator ITexture*() const { return AsTexture(); }
        operator IBuffer*() const { return AsBuffer(); }
        IDeviceObject* operator->() const { return m_r; }

    private:
        ITextureView* View(TEXTURE_VIEW_TYPE t) const
        {
            ITexture* tex = AsTexture();
            if (!tex) { Recorder::Get().Error("ResourceRegistry: a view of something that is not a texture"); return nullptr; }
            return tex->GetDefaultView(t);
        }
        RefCntAutoPtr<IDeviceObject>& m_r;
    };
    ResourceAccessor operator[](ResourceIdType id) const { return ResourceAccessor{const_cast<RefCntAutoPtr<IDeviceObject>&>(m_Resources[id])}; }
    void             Clear() { for (auto& r : m_Resources) r.Release(); }

private:
    std::vector<RefCntAutoPtr<IDeviceObject>> m_Resources;
};

// ---- ShaderMacroHelper.hpp
class ShaderMacroHelper
{
public:
    ShaderMacroHelper() = default;
    ShaderMacroHelper(std::initializer_list<ShaderMacro> init)
    {
        for (const ShaderMacro& x : init) Put(x.Name, x.Definition ? x.Definition : "");
    }
    ShaderMacroHelper(const ShaderMacroHelper& o) : m_Items{o.m_Items} {}
    ShaderMacroHelper& operator=(const ShaderMacroHelper& o)
    {
        m_Items = o.m_Items;
        return *this;
    }
    ShaderMacroHelper& Add(const Char* name, const Char* def) { return Put(name, def ? def : ""); }
    ShaderMacroHelper& Add(const Char* name, bool v) { return Put(name, v ? "1" : "0"); }
    ShaderMacroHelper& Add(const Char* name, int v) { return Put(name, std::to_string(v)); }
    ShaderMacroHelper& Add(const Char* name, Uint32 v) { return Put(name, std::to_string(v) + "u"); }
    ShaderMacroHelper& Add(const Char* name, float v)
    {
        char b[64];
        std::snprintf(b, sizeof(b), "%.9g", v);
        return Put(name, b);
    }
    template <class T, class = std::enable_if_t<std::is_enum_v<T>>> ShaderMacroHelper& Add(const Char* name, T v) { return Add(name, static_cast<int>(v)); }
    void Clear() { m_Items.clear(); }
    operator ShaderMacroArray() const
    {
        m_Array.clear();
        for (const auto& kv : m_Items) m_Array.push_back(ShaderMacro{kv.first.c_str(), kv.second.c_str()});
        return ShaderMacroArray{m_Array.data(), Uint32(m_Array.size())};
    }

private:
    ShaderMacroHelper& Put(const Char* name, const std::string& def)
    {
        for (auto& kv : m_Items)
            if (kv.first == name) { kv.second = def; return *this; }
        m_Items.emplace_back(name, def);
        return *this;
    }
    std::vector<std::pair<std::string, std::string>> m_Items;
    mutable std::vector<ShaderMacro>                 m_Array;
};

// ---- GraphicsTypesX.hpp
class PipelineResourceLayoutDescX
{
public:
    PipelineResourceLayoutDescX& AddVariable(SHADER_TYPE stages, const Char* name, SHADER_RESOURCE_VARIABLE_TYPE type, SHADER_VARIABLE_FLAGS flags = SHADER_VARIABLE_FLAG_NONE)
    {
        m_Names.push_back(std::make_unique<std::string>(name));
        m_Vars.push_back(ShaderResourceVariableDesc{m_Names.back()->c_str(), stages, type, flags});
        return *this;
    }
    PipelineResourceLayoutDescX& AddImmutableSampler(SHADER_TYPE stages, const Char* name, const SamplerDesc& desc)
    {
        m_Names.push_back(std::make_unique<std::string>(name));
        m_Sams.push_back(ImmutableSamplerDesc{stages, m_Names.back()->c_str(), desc});
        return *this;
    }
    PipelineResourceLayoutDescX& SetDefaultVariableType(SHADER_RESOURCE_VARIABLE_TYPE t)
    {
        m_Default = t;
        return *this;
    }
    PipelineResourceLayoutDescX& SetDefaultVariableMergeStages(SHADER_TYPE) { return *this; } // (the stand-in keeps one variable table per pipeline: stages are always merged)
    operator PipelineResourceLayoutDesc() const
    {
        PipelineResourceLayoutDesc d;
        d.DefaultVariableType  = m_Default;
        d.NumVariables         = Uint32(m_Vars.size());
        d.Variables            = m_Vars.data();
        d.NumImmutableSamplers = Uint32(m_Sams.size());
        d.ImmutableSamplers    = m_Sams.data();
        return d;
    }

private:
    SHADER_RESOURCE_VARIABLE_TYPE             m_Default = SHADER_RESOURCE_VARIABLE_TYPE_STATIC;
    std::vector<std::unique_ptr<std::string>> m_Names;
    std::vector<ShaderResourceVariableDesc>   m_Vars;
    std::vector<ImmutableSamplerDesc>         m_Sams;
};
class ShaderResourceVariableX
{
public:
    ShaderResourceVariableX(IPipelineState* pso, SHADER_TYPE t, const Char* name) : m_Name{name}
    {
        m_pVar = pso ? pso->GetStaticVariableByName(t, name) : nullptr;
        if (!m_pVar) Recorder::Get().Error(std::string("ShaderResourceVariableX: no STATIC variable '") + name + "' in pipeline '" + (pso ? pso->name : "null") + "'");
    }
    ShaderResourceVariableX(IShaderResourceBinding* srb, SHADER_TYPE t, const Char* name) : m_Name{name}
    {
        m_pVar = srb ? srb->GetVariableByName(t, name) : nullptr;
        // (a name the resource layout does not list is a variable of the default type -- static -- or one the shader permutation does not have: the reference sets
        //  g_TextureMotion of the ray march whether or not SSR_OPTION_PREVIOUS_FRAME compiled it in; such a Set() is a no-op, noted in the list)
        if (!m_pVar) Recorder::Get().Emit(std::string("{\"op\":\"note\",\"what\":\"no mutable / dynamic variable '") + name + "' in a binding of pipeline '" + (srb && srb->pso ? srb->pso->name : "null") + "'\"}");
    }
    void Set(IDeviceObject* obj, SET_SHADER_RESOURCE_FLAGS f = SET_SHADER_RESOURCE_FLAG_NONE)
    {
        if (m_pVar) m_pVar->Set(obj, f);
    }
    void SetArray(IDeviceObject* const* objs, Uint32 first, Uint32 n, SET_SHADER_RESOURCE_FLAGS f = SET_SHADER_RESOURCE_FLAG_NONE)
    {
        if (m_pVar) m_pVar->SetArray(objs, first, n, f);
    }
    explicit operator bool() const { return m_pVar != nullptr; }

private:
    IShaderResourceVariable* m_pVar = nullptr;
    std::string              m_Name;
};

// ---- ScopedDebugGroup.hpp
class ScopedDebugGroup
{
public:
    ScopedDebugGroup(IDeviceContext* ctx, const Char* name, const float* color = nullptr) : m_pCtx{ctx} { m_pCtx->BeginDebugGroup(name, color); }
    ~ScopedDebugGroup() { m_pCtx->EndDebugGroup(); }
    ScopedDebugGroup(const ScopedDebugGroup&) = delete;

private:
    IDeviceContext* m_pCtx;
};

// ---- MapHelper.hpp
template <class T, bool KeepStrong = false> class MapHelper
{
public:
    MapHelper() = default;
    MapHelper(IDeviceContext* ctx, IBuffer* buf, MAP_TYPE type, MAP_FLAGS flags) { Map(ctx, buf, type, flags); }
    ~MapHelper() { Unmap(); }
    void Map(IDeviceContext* ctx, IBuffer* buf, MAP_TYPE type, MAP_FLAGS flags)
    {
        m_pCtx = ctx; m_pBuf = buf; m_Type = type;
        void* p = nullptr;
        if (buf) ctx->MapBuffer(buf, type, flags, p);
        m_p = static_cast<T*>(p);
    }
    void Unmap()
    {
        if (m_p && m_pBuf) m_pCtx->UnmapBuffer(m_pBuf, m_Type);
        m_p = nullptr;
    }
    operator T*() { return m_p; }
    explicit operator bool() const { return m_p != nullptr; }
    T*       operator->() { return m_p; }
    T&       operator[](size_t i) { return m_p[i]; }
    T&       operator*() { return *m_p; }

private:
    IDeviceContext* m_pCtx = nullptr;
    IBuffer*        m_pBuf = nullptr;
    MAP_TYPE        m_Type = MAP_WRITE;
    T*              m_p    = nullptr;
};

// ---- GraphicsUtilities.h
inline Uint32 ComputeMipLevelsCount(Uint32 w, Uint32 h = 0, Uint32 d = 0)
{
    Uint32 s = std::max(std::max(w, h), d);
    if (s == 0) return 0;
    Uint32 m = 0;
    while ((s >> m) > 0) ++m;
    return m;
}
inline void CreateUniformBuffer(IRenderDevice* pDevice, Uint64 size, const Char* name, IBuffer** ppBuffer, USAGE usage = USAGE_DYNAMIC, BIND_FLAGS bind = BIND_UNIFORM_BUFFER,
                                CPU_ACCESS_FLAGS cpu = CPU_ACCESS_WRITE, void* pInitialData = nullptr)
{
    BufferDesc d;
    d.Name = name; d.Size = size; d.Usage = usage; d.BindFlags = bind; d.CPUAccessFlags = cpu;
    BufferData init{pInitialData, size};
    pDevice->CreateBuffer(d, pInitialData ? &init : nullptr, ppBuffer);
}
template <class T> inline void CreateUniformBuffer(IRenderDevice* pDevice, Uint64 size, const Char* name, T&& pp, USAGE usage = USAGE_DYNAMIC, BIND_FLAGS bind = BIND_UNIFORM_BUFFER,
                                                   CPU_ACCESS_FLAGS cpu = CPU_ACCESS_WRITE, void* pInitialData = nullptr)
{
    IBuffer** raw = pp;
    CreateUniformBuffer(pDevice, size, name, raw, usage, bind, cpu, pInitialData);
}

// ---- GraphicsTypesX.hpp: GraphicsPipelineStateCreateInfoX (what Components/src/EnvMapRenderer.cpp uses of it)
struct GraphicsPipelineStateCreateInfoX : GraphicsPipelineStateCreateInfo
{
    explicit GraphicsPipelineStateCreateInfoX(const char* name = nullptr)
    {
        m_Name = name ? name : "";
        PSODesc.Name = m_Name.c_str();
    }
    GraphicsPipelineStateCreateInfoX& SetResourceLayout(const PipelineResourceLayoutDescX& l)
    {
        PSODesc.ResourceLayout = l; // (the arrays belong to `l`, which outlives the pipeline creation in every caller)
        return *this;
    }
    GraphicsPipelineStateCreateInfoX& AddShader(IShader* s)
    {
        if (s && s->type == SHADER_TYPE_VERTEX) pVS = s;
        else pPS = s;
        return *this;
    }
    GraphicsPipelineStateCreateInfoX& SetPrimitiveTopology(PRIMITIVE_TOPOLOGY t) { GraphicsPipeline.PrimitiveTopology = t; return *this; }
    GraphicsPipelineStateCreateInfoX& SetDepthFormat(TEXTURE_FORMAT f) { GraphicsPipeline.DSVFormat = f; return *this; }
    GraphicsPipelineStateCreateInfoX& AddRenderTarget(TEXTURE_FORMAT f)
    {
        GraphicsPipeline.RTVFormats[GraphicsPipeline.NumRenderTargets++] = f;
        return *this;
    }

private:
    std::string m_Name;
};

// ---- ShaderSourceFactoryUtils.hpp: the factories only travel into ShaderCreateInfo; the stand-in does not open shader files
struct MemoryShaderSourceFileInfo
{
    const Char* Name = nullptr;
    const Char* pData = nullptr;
    MemoryShaderSourceFileInfo() = default;
    MemoryShaderSourceFileInfo(const Char* n, const Char* d) : Name{n}, pData{d} {}
};
inline RefCntAutoPtr<IShaderSourceInputStreamFactory> CreateMemoryShaderSourceFactory(std::initializer_list<MemoryShaderSourceFileInfo>)
{
    return RefCntAutoPtr<IShaderSourceInputStreamFactory>{new IShaderSourceInputStreamFactory()};
}
inline RefCntAutoPtr<IShaderSourceInputStreamFactory> CreateCompoundShaderSourceFactory(std::initializer_list<IShaderSourceInputStreamFactory*>)
{
    return RefCntAutoPtr<IShaderSourceInputStreamFactory>{new IShaderSourceInputStreamFactory()};
}

// ---- CommonlyUsedStates.h
static const DepthStencilStateDesc DSS_DisableDepth{false, false};
static const DepthStencilStateDesc DSS_EnableDepthNoWrites{true, false};
static const BlendStateDesc        BS_Default{};
static const SamplerDesc           Sam_LinearClamp{FILTER_TYPE_LINEAR, FILTER_TYPE_LINEAR, FILTER_TYPE_LINEAR, TEXTURE_ADDRESS_CLAMP, TEXTURE_ADDRESS_CLAMP, TEXTURE_ADDRESS_CLAMP};
static const SamplerDesc           Sam_PointClamp{FILTER_TYPE_POINT, FILTER_TYPE_POINT, FILTER_TYPE_POINT, TEXTURE_ADDRESS_CLAMP, TEXTURE_ADDRESS_CLAMP, TEXTURE_ADDRESS_CLAMP};
static const SamplerDesc           Sam_PointWrap{FILTER_TYPE_POINT, FILTER_TYPE_POINT, FILTER_TYPE_POINT, TEXTURE_ADDRESS_WRAP, TEXTURE_ADDRESS_WRAP, TEXTURE_ADDRESS_WRAP};
static const SamplerDesc           Sam_LinearWrap{FILTER_TYPE_LINEAR, FILTER_TYPE_LINEAR, FILTER_TYPE_LINEAR, TEXTURE_ADDRESS_WRAP, TEXTURE_ADDRESS_WRAP, TEXTURE_ADDRESS_WRAP};
} // namespace Diligent

// ---- imgui.h / ImGuiUtils.hpp: nothing is drawn
enum ImGuiSliderFlags_ { ImGuiSliderFlags_None = 0, ImGuiSliderFlags_AlwaysClamp = 16, ImGuiSliderFlags_Logarithmic = 32 };
namespace ImGui
{
inline bool Combo(const char*, int*, const char* const[], int, int = -1) { return false; }
inline bool BeginCombo(const char*, const char*, int = 0) { return false; }
inline void EndCombo() {}
inline bool Selectable(const char*, bool = false, int = 0) { return false; }
inline void SetItemDefaultFocus() {}
inline bool Checkbox(const char*, bool*) { return false; }

inline bool SliderFloat(const char*, float*, float, float, const char* = "%.3f", int = 0) { return false; }
inline bool SliderInt(const char*, int*, int, int, const char* = "%d", int = 0) { return false; }
inline void Spacing() {}
inline void TextDisabled(const char*, ...) {}
inline void Text(const char*, ...) {}
inline void HelpMarker(const char*, bool = true, const char* = "(?)") {}
struct ScopedDisabler
{
    explicit ScopedDisabler(bool, float = 0.25f) {}
};
} // namespace ImGui
