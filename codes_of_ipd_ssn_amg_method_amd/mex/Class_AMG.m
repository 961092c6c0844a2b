function [x,it,rel_res,rel_resk,rhok] = Class_AMG(varargin)
% Drop-in shim with the reference's signature: forwards to libipdamg (HIP, gfx950)
% through the MEX gateway ipd_mex.  See INTEGRATION.md.
[x,it,rel_res,rel_resk,rhok] = ipd_mex('Class_AMG', varargin{:});
end
