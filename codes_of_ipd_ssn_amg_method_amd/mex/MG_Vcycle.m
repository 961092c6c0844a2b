function e = MG_Vcycle(varargin)
% Drop-in shim with the reference's signature: forwards to libipdamg (HIP, gfx950)
% through the MEX gateway ipd_mex.  See INTEGRATION.md.
[e] = ipd_mex('MG_Vcycle', varargin{:});
end
